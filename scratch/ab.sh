#!/bin/bash
# A/B of library variants on the GPU box:  bash scratch/ab.sh "v0 v1 ..." [extra bench args]
# each scratch/variants/<name>.so is copied over envgs_amd/libenvgs_hip.so and the default bench workload is run twice (second run reported too)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VARS="$1"; shift
cp envgs_amd/libenvgs_hip.so /tmp/_orig.so
for v in $VARS; do
  cp scratch/variants/$v.so envgs_amd/libenvgs_hip.so
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 "$@" 2>/dev/null | python scratch/ab_show.py "$v.$rep"
  done
done
cp /tmp/_orig.so envgs_amd/libenvgs_hip.so
