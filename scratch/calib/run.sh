#!/bin/bash
# GPU box:  bash scratch/calib/run.sh   -> per-kernel FETCH_SIZE (KB) of the calibration patterns next to their known byte counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib
rm -rf $O; mkdir -p $O
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o c -- $R/scratch/calib/fetch_calib > $O/stdout.txt 2>&1
cat $O/stdout.txt | grep expected
python - <<'P'
import csv, os, collections, re
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/calib/c_counter_collection.csv")
d = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    if r["Counter_Name"] == "FETCH_SIZE": d[re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))].append(float(r["Counter_Value"]))
for k, v in d.items(): print("%-22s launches %d  FETCH_SIZE avg %.0f KB = %.3f GB" % (k, len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e9))
P
