cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_trace_parity.py tests/test_envgs_step_parity.py -x -q -m gpu -k "defer or deferred or barrier" 2>&1 | tail -12
python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envgs', d['ms_per_step'], d['value'])"
