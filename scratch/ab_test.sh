#!/bin/bash
# parity tests of the tracer under a library VARIANT, then the A/B:  bash scratch/ab_test.sh <variant> "<ab list>"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp envgs_amd/libenvgs_hip.so /tmp/_orig_t.so; cp scratch/variants/$1.so envgs_amd/libenvgs_hip.so
python -m pytest tests/test_trace_parity.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed" | tail -2
cp /tmp/_orig_t.so envgs_amd/libenvgs_hip.so
bash scratch/ab.sh "$2" 2>&1 | grep -v "^  "
