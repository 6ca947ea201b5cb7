#!/bin/bash
# forward-only timing of library variants:  bash scratch/ab_rh.sh "v0 v1 ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp envgs_amd/libenvgs_hip.so /tmp/_orig.so
for v in $1; do
  cp scratch/variants/$v.so envgs_amd/libenvgs_hip.so
  for rep in 1 2; do timeout 200 python scratch/rh_time.py $v.$rep 2>&1 | grep "^$v" ; done
done
cp /tmp/_orig.so envgs_amd/libenvgs_hip.so
