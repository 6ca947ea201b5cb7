#!/bin/bash
# default bench with the list capacity pinned (HIT_CAP force) vs adaptive: what the long-list pass costs when cap flickers across 256
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for c in 0 256 320 0 256 320; do
  python -c "
import sys; sys.argv=['bench.py','--no-cpu-baseline','--no-render','--no-reference-caller','--steps','20','--warmup','5']
from envgs_amd import tracing
if $c: tracing.HIT_CAP['force']=$c
import bench; bench.main()" 2>/dev/null | python scratch/ab_show.py "cap=$c" | head -1
done
