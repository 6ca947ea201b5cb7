cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_trace_parity.py -x -q -m gpu -k "deferred" 2>&1 | tail -2
