"""Kernel inventory of ONE step from a rocprofv3 kernel trace:  python scratch/step_inventory.py <kernel_trace.csv> [max_us]
prints the big kernels on a timeline and the small ones (< max_us, default 30) grouped by name -- what a step spends on launches."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "fused_adam" in r["Kernel_Name"]]
step = rows[adam[-3] + 1:adam[-2] + 1]
t0 = int(step[0]["Start_Timestamp"])
nm = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")).replace("envgs::", "").replace("at::native::", "at::")[:80]
print("step: %d kernels, %.3f ms from first start to last end" % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e6))
c = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= lim: print("%8.3f +%8.1f us  q%s %s" % (s, d, r["Queue_Id"], nm(r)))
    else: c[nm(r)][0] += 1; c[nm(r)][1] += d
print("small kernels: %d, %.0f us" % (sum(v[0] for v in c.values()), sum(v[1] for v in c.values())))
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1]): print("%3d x %6.1f us  %s" % (v[0], v[1], k))
