"""Reduce rocprofv3 counter_collection.csv files in place to per-(kernel, counter) sums and launch counts -- the raw files hold one row per
launch, counter and XCD/SE instance (hundreds of MB for the configs[4] workload) and gpurun merges at most 64 MiB back.
    python profiles/shrink_counters.py <dir> [<dir> ...]"""
import collections, csv, glob, os, re, sys
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, set()])
        with open(path) as f:
            for r in csv.DictReader(f):
                k = (re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")), r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        with open(path, "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])     # Counter_Value = average per launch
            for (kn, cn), (tot, ids) in sorted(acc.items()):
                w.writerow([kn, cn, tot / max(len(ids), 1), len(ids)])
