#!/bin/bash
# kernel-level durations of the raster-only workload:  bash scratch/prof_raster.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/p_rb
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_rb -o rb -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --workload raster --steps 30 --warmup 4 "$@" > /dev/null 2>&1
python - <<'P'
import csv, os, re
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/p_rb/rb_kernel_stats.csv"))))
for r in rows:
    if "envgs::" in r["Name"]:
        print("%-40s calls %4s avg %8.1f us" % (re.sub(r"\(.*", "", r["Name"].replace("void ", ""))[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
P
