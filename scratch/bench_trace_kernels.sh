timeout 150 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('trace_counts')); [print(k, v['ms'], v['launches']) for k,v in d['kernels'].items() if k.startswith('trace')]"
