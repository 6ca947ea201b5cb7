#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for sg in 2 3 4; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-segments $sg 2>/dev/null | python scratch/ab_show.py "seg=$sg" | head -1
done
for w in 3 5; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-segments 3 --debug-collect-wgs $w 2>/dev/null | python scratch/ab_show.py "seg=3,wgs=$w" | head -1
done
