#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for wg in 0 3 4 5 6; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --repeats 3 --debug-collect-wgs $wg 2>/dev/null | python scratch/ab_show.py "wg=$wg.$rep" | head -1
done; done
