#!/bin/bash
# A/B of ENVGS_DBG_TRACE values on ONE library, interleaved:  bash scratch/ab_switch2.sh "0 32" [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VALS="$1"; shift
for rep in 1 2; do for v in $VALS; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-trace $v "$@" 2>/dev/null | python scratch/ab_show.py "trace=$v.$rep"
done; done
