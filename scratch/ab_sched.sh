#!/bin/bash
# round 5: forward schedule sweep -- segments x collection workgroups per CU (0 = the shipped default: 2 segments, 4 + 5 workgroups)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for seg in 0 1 3; do for wg in 0 3 5 6 8; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --repeats 3 --debug-segments $seg --debug-collect-wgs $wg 2>/dev/null | python scratch/ab_show.py "seg=$seg.wg=$wg" | head -1
done; done
