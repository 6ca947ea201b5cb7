#!/bin/bash
# Round 6: A/B of batch_surfel_bwd library variants on the GPU box:  bash scratch/ab_bsb.sh "v0 v1 ..."   (scratch/bsb_ab.py under each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp envgs_amd/libenvgs_hip.so /tmp/_orig.so
for v in $1; do
  cp scratch/variants/$v.so envgs_amd/libenvgs_hip.so
  timeout 300 python scratch/bsb_ab.py $v 2>&1 | grep -v Warning | tail -12
done
cp /tmp/_orig.so envgs_amd/libenvgs_hip.so
