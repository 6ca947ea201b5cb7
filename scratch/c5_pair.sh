#!/bin/bash
# configs[4] bench, fp16 vs fp32 feature storage: per-kernel HIP-event times side by side
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for dt in f16 f32; do
  timeout 250 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype $dt --no-cpu-baseline --no-reference-caller --no-render --steps 8 --warmup 3 --repeats 3 "$@" 2>/dev/null > gpurun_out/c5_$dt.json
  python - <<PY
import json
d = json.load(open("gpurun_out/c5_$dt.json"))
print("$dt", d["ms_per_step"], " ".join("%s=%.3fx%d" % (k.replace("trace.", ""), v["ms"], v["launches"]) for k, v in d["kernels"].items() if v["ms"] > 0.2))
PY
done
