#!/bin/bash
# quick check of a tracer change: the trace parity tests, then two short bench runs (kernel times from the prof layer)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_trace_parity.py tests/test_bvh_structure.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 30 --warmup 6 2>/dev/null | python scratch/ab_show.py "run$i"
done
