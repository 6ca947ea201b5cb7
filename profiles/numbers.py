"""Markdown table of a round's bench lines (profiles/<tag>_bench_*.json):  python profiles/numbers.py r06"""
import glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.abspath(__file__))
print("| run (profiles/%s_bench_<name>.json) | ms / step (min-max of the timed regions) | it/s | dominant kernel: ms x launches, GB/s, frac of 8 TB/s, VALU / measured peak | notes |" % tag)
print("|---|---|---|---|---|")
for p in sorted(glob.glob(os.path.join(root, "%s_bench_*.json" % tag))):
    try:
        d = json.load(open(p))
    except Exception as e:
        print("| %s | unreadable: %s | | | |" % (os.path.basename(p), e)); continue
    r = d.get("roofline") or {}
    k = (d.get("kernels") or {}).get(r.get("kernel"), {})
    per = max(1, round(k.get("launches", 0) / max(d["steps"] * d["ms_per_step_spread"]["repeats"], 1))) if k else 0
    iss = (r.get("issue") or {}).get("valu_util_measured_peak")
    ks = d.get("kernels") or {}
    extra = []
    for n in ("trace.collect_hits", "trace.sort_composite_fwd", "trace.register_hits", "trace.batch_surfel_bwd", "trace.reduce_surfel_records", "composite_fwd", "composite_bwd"):
        if n in ks and n != r.get("kernel"):
            extra.append("%s %.3f" % (n.replace("trace.", ""), ks[n]["ms"]))
    cfg = d["config"]
    note = "; ".join(extra)
    if cfg.get("reference_caller_ms_per_step"): note += "; unchanged-caller form %.2f ms (%s)" % (cfg["reference_caller_ms_per_step"], cfg.get("reference_caller_ms_by_torch_blas"))
    if d.get("render_ms_per_view"): note += "; render %.2f ms / view = %.0f Mpix/s" % (d["render_ms_per_view"], d["render_mpix_per_s"])
    lo = ((r.get("raster_composite_bwd") or {}).get("lane_occupancy") or {}).get("lanes_per_pass")
    if lo: note += "; R7 lanes / pass %.2f" % lo
    print("| %s | %.2f (%.2f-%.2f) | %.1f | %s: %.3f x %d, %s GB/s, %s, %s | %s |" % (
        os.path.basename(p).replace(tag + "_bench_", "").replace(".json", ""), d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["max"], d["value"],
        r.get("kernel"), r.get("ms_per_launch", 0), per, r.get("achieved"), r.get("frac"), iss, note))
