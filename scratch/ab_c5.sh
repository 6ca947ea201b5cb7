#!/bin/bash
# configs[4] bench under library variants: per-kernel HIP-event times   bash scratch/ab_c5.sh "v0 v1 ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp envgs_amd/libenvgs_hip.so /tmp/_orig.so
for v in $1; do
  cp scratch/variants/$v.so envgs_amd/libenvgs_hip.so
  timeout 250 python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-reference-caller --no-render --steps 8 --warmup 3 --repeats 3 2>/dev/null > /tmp/c5_$v.json
  python - <<PY
import json
d = json.load(open("/tmp/c5_$v.json"))
print("$v", d["ms_per_step"], " ".join("%s=%.3fx%d" % (k.replace("trace.", ""), v["ms"], v["launches"]) for k, v in d["kernels"].items() if v["ms"] > 0.2))
PY
done
cp /tmp/_orig.so envgs_amd/libenvgs_hip.so
