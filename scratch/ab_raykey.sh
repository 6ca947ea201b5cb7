#!/bin/bash
# A/B of the ray coherence key layouts on ONE library:  bash scratch/ab_raykey.sh "9 4 3" [extra bench args]   (value - 1 = direction-only lead rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VALS="$1"; shift
for v in $VALS; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-raykey $v "$@" 2>/dev/null | python scratch/ab_show.py "raykey=$v"
done
