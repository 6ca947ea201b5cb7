# Run on the GPU box AFTER profiles/collect_profiles.sh + `python profiles/summarize.py r06` (the lines quote the counters of their own workload):
#   bash profiles/final_benches.sh    -> gpurun_out/bench_<name>_final.json, copied to profiles/r06_bench_<name>.json by summarize.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
Q="--no-cpu-baseline --no-reference-caller"
timeout 500 python bench.py > $O/bench_envgs_final.json 2> $O/bench_envgs_final.err
timeout 150 python bench.py --caller reference --no-cpu-baseline > $O/bench_envgs_reference_caller_final.json 2> $O/bench_envgs_reference_caller_final.err
timeout 150 python bench.py --caller twin $Q > $O/bench_envgs_twin_caller_final.json 2> $O/bench_envgs_twin_caller_final.err
timeout 150 python bench.py --no-deferred-surfel-grads $Q > $O/bench_envgs_stream_ordered_final.json 2> $O/bench_envgs_stream_ordered_final.err
timeout 250 python bench.py --workload raster > $O/bench_raster_final.json 2> $O/bench_raster_final.err
timeout 150 python bench.py --env-gaussians 700000 $Q --steps 15 --warmup 4 > $O/bench_env700k_final.json 2> $O/bench_env700k_final.err
timeout 200 python bench.py --gaussians 1800000 --env-gaussians 630000 $Q --steps 10 --warmup 3 > $O/bench_caps_final.json 2> $O/bench_caps_final.err
timeout 150 python bench.py --feature-dtype f16 $Q > $O/bench_envgs_f16_final.json 2> $O/bench_envgs_f16_final.err
timeout 300 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 $Q --steps 8 --warmup 3 --step-times 8 > $O/bench_config5_final.json 2> $O/bench_config5_final.err
timeout 300 python bench.py --workload base_trace --trace-depth 0 --no-reference-caller --cpu-rays 1024 > $O/bench_base_trace_d0_final.json 2> $O/bench_base_trace_d0_final.err
timeout 300 python bench.py --workload base_trace --trace-depth 2 $Q --steps 10 --warmup 3 > $O/bench_base_trace_d2_final.json 2> $O/bench_base_trace_d2_final.err
bash scratch/bvh_prof.sh > $O/bvh_prof.txt 2>&1
python -m pytest tests -q -m perf -s > $O/perf_final.log 2>&1
for f in $O/bench_*_final.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-52s %8.3f ms/step %9.2f it/s  roofline %s frac %s  issue %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], (d["roofline"] or {}).get("kernel"), (d["roofline"] or {}).get("frac"),
          ((d["roofline"] or {}).get("issue") or {}).get("valu_util_measured_peak")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
