#!/bin/bash
# A/B of forward_prepare beside the coherence sort (default) against after it (--debug-trace 16384), alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_trace_parity.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do
  for dt in 0 16384; do
    python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --debug-trace $dt 2>/dev/null | python scratch/ab_show.py "dt$dt.$rep"
  done
done
python bench.py --no-cpu-baseline --no-render --no-reference-caller --workload envgs --channels 7 --trace-depth 2 --steps 10 --warmup 3 2>/dev/null | python scratch/ab_show.py "c5.new"
python bench.py --no-cpu-baseline --no-render --no-reference-caller --workload envgs --channels 7 --trace-depth 2 --steps 10 --warmup 3 --debug-trace 16384 2>/dev/null | python scratch/ab_show.py "c5.old"
