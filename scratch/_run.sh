cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b_two.json 2> gpurun_out/b_two.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_two.json').read().strip().splitlines()[-1])
r=d["roofline"]
print(d["ms_per_step"], d["value"], r["kernel"], r["ms_per_launch"], r["frac"], r.get("kernel_timing"))
print([(k["kernel"], k["ms_per_launch"], k["launches_per_step"]) for k in r["next_kernels"]], r["raster_composite_bwd"]["ms_per_launch"], d["config"]["extension_ms_per_step"])
PY
for m in two all none; do python bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 5 --live-scopes $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'])"; done
python bench.py --no-cpu-baseline --no-render --workload raster --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('raster', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['ms_per_launch'], d['roofline'].get('kernel_timing',{}).get('in_the_timed_regions'))"
timeout 300 python -m pytest tests/test_bench_two_ranks.py -x -q -m gpu 2>&1 | tail -1
