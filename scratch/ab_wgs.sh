#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in $1; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-collect-wgs $v 2>/dev/null | python scratch/ab_show.py "wgs=$v" | head -1
done
