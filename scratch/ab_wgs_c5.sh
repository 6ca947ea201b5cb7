#!/bin/bash
# configs[4] / base_trace under --debug-collect-wgs values:  bash scratch/ab_wgs_c5.sh "0 4"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in $1; do
  timeout 250 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-reference-caller --no-render --steps 8 --warmup 3 --repeats 3 --debug-collect-wgs $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config5 wgs=$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k in ('trace_fwd','trace.collect_hits','trace.sort_composite_fwd','trace.register_hits')})"
  timeout 250 python bench.py --workload base_trace --no-cpu-baseline --no-render --steps 20 --warmup 5 --repeats 3 --debug-collect-wgs $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('base_trace_d0 wgs=$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k in ('trace_fwd','trace.collect_hits','trace.sort_composite_fwd','trace.register_hits')})"
done
