"""Turn the rocprofv3 outputs merged under gpurun_out/ into the small tracked summaries in profiles/ (run after a profiling gpurun call)."""
import collections, csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"


def short(n):
    m = re.search(r'radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|init_lookback_scan_state|radix_sort_block_sort|radix_sort_merge', n)
    if m: return 'rocprim::' + m.group(0)
    n = re.sub(r'\(.*', '', n.replace('void ', '')).replace('at::native::', 'at::')
    return n[:90].replace(',', ';')


def stats(tag, steps, desc):
    path = os.path.join(ROOT, 'gpurun_out', 'p_%s' % tag, '%s_kernel_stats.csv' % tag)
    if not os.path.exists(path): return
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    with open(os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.csv' % (TAG, tag)), 'w') as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv (MI355X)  -- %s\n" % desc)
        f.write("# %d steps incl. warm-up; sum of all kernel time = %.3f ms per step\n" % (steps, tot / steps / 1e6))
        f.write("name,calls,total_ns,avg_ns,pct,ms_per_step\n")
        for r in rows[:32]:
            f.write("%s,%s,%s,%.0f,%s,%.4f\n" % (short(r['Name']), r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['Percentage'], float(r['TotalDurationNs']) / steps / 1e6))


def pmc():
    def load(path, cname):
        d = collections.defaultdict(list)
        if not os.path.exists(path): return d
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == cname:
                d[re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))].append(float(r['Counter_Value']))
        return d
    f = load(os.path.join(ROOT, 'gpurun_out/pmc_fetch/f_counter_collection.csv'), 'FETCH_SIZE')
    w = load(os.path.join(ROOT, 'gpurun_out/pmc_write/w_counter_collection.csv'), 'WRITE_SIZE')
    names = {'envgs::composite_fwd': 'composite_fwd', 'envgs::composite_bwd': 'composite_bwd', 'envgs::project_surfels': 'project_surfels',
             'envgs::project_surfels_bwd': 'project_surfels_bwd', 'envgs::bin_pass': 'bin_tile_pairs', 'envgs::sort_tile_lists': 'sort_tile_lists',
             'envgs::collect_hits_coop': 'trace.collect_hits', 'envgs::collect_hits_packet4': 'trace.collect_hits(one wavefront per batch)', 'envgs::collect_hits_packet': 'trace.collect_hits(binary)', 'envgs::collect_hits': 'trace.collect_hits(per-ray)', 'envgs::sort_composite_fwd<4, false, true>': 'trace.sort_composite_fwd', 'envgs::sort_composite_fwd<4, false, false>': 'trace.sort_composite_fwd(per-lane SH gathers)',
             'envgs::register_hits': 'trace.register_hits', 'envgs::batch_surfel_bwd': 'trace.batch_surfel_bwd',
             'envgs::reduce_surfel_records': 'trace.reduce_surfel_records'}
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (two separate passes, no other trace domains) -- "
                   "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render on MI355X; per-launch averages. Units per MI355X_MICROARCH.md: counters are KB; on gfx950 "
                   "FETCH_SIZE reports 1/2 of the bytes of 16 B/lane reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. WRITE_SIZE is uncalibrated "
                   "(scattered 4-8 B stores are counted at 32 B granularity). Calibrated in round 3 (profiles/r03_fetch_calibration.txt): FETCH_SIZE = read requests x 64 B -- a coalesced "
                   "stream issues 128 B requests (hence the x2), a scattered 16 / 32 B gather ONE request per lane-access, so for the gather-dominated tracer kernels "
                   "(sort_composite_fwd, register_hits, batch_surfel_bwd) hbm_bytes over-states the reads by up to 2x; fetch_requests = FETCH_SIZE*1024/64 is exact for every kernel.", "kernels": {}}
    def pick(d, k):                         # exact name, or any template instantiation of it (kernel<...>), launches pooled
        vals = []
        for name, v in d.items():
            if name == k or name.startswith(k + '<'):
                vals += v
        return vals
    for k, v in names.items():
        fv_, wv_ = pick(f, k), pick(w, k)
        if fv_:
            fv = sum(fv_) / len(fv_); wv = sum(wv_) / len(wv_) if wv_ else 0
            out["kernels"][v] = {"FETCH_SIZE_KB": round(fv, 1), "WRITE_SIZE_KB": round(wv, 1), "hbm_bytes": int((2 * fv + wv) * 1024), "fetch_requests": int(fv * 1024 / 64)}
    # every other counter pass (gpurun_out/pmc_<anything>/**/*counter_collection.csv): per-launch averages per kernel
    import glob
    other = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                continue
            other[re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in names.items():
        agg = collections.defaultdict(list)
        for name, ctrs in other.items():
            if name == k or name.startswith(k + '<'):
                for c, vals in ctrs.items():
                    agg[c] += vals
        if agg:
            out["kernels"].setdefault(v, {})
            for c, vals in sorted(agg.items()):
                out["kernels"][v][c] = round(sum(vals) / len(vals), 1)
            out["kernels"][v]["launches_sampled"] = len(next(iter(agg.values())))
    out["_sq"] = ("SQ_* = per-launch averages of separate `rocprofv3 --pmc <4-8 SQ counters> --kernel-trace` passes over the same command "
                  "(profiles/collect_profiles.sh); SQ_INSTS_* count wave-level instructions, SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count "
                  "quad-cycles, SQ_BUSY_CU_CYCLES counts cycles summed over CUs (MI355X_MICROARCH.md, rocprofv3 PMC slots)")
    if out["kernels"]:
        json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_pmc_envgs.json' % TAG), 'w'), indent=1)


stats('envgs', 204, 'full EnvGS step: 300k base surfels ch05 raster + 163840 env surfels LBVH trace, 800x800; python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-render --no-reference-caller (10 timed regions of 20 steps + 4 warm-up steps) (the tracer forward runs as two batch segments on two streams: 2 launches per step, overlapping, so the kernel times sum to more than the step)')
stats('raster', 304, 'raster only: 300k surfels, SH deg 3 in-kernel, 800x800; python bench.py --workload raster --steps 30 --warmup 4 --no-cpu-baseline --no-render (10 timed regions of 30 steps + 4 warm-up steps)')
pmc()
for src_, dst_ in (("step_inventory.txt", "step_inventory.txt"),):
    sp = os.path.join(ROOT, 'gpurun_out', src_)
    if os.path.exists(sp) and os.path.getsize(sp) > 10:
        open(os.path.join(ROOT, 'profiles', '%s_%s' % (TAG, dst_)), 'w').write(open(sp).read())
for P_ in (163840, 700000):
    sp = os.path.join(ROOT, 'gpurun_out', 'p_bvh', 'bvh%d_kernel_stats.csv' % P_)
    if os.path.exists(sp):
        rows = list(csv.DictReader(open(sp)))
        with open(os.path.join(ROOT, 'profiles', '%s_bvh_%d_kernel_stats.csv' % (TAG, P_)), 'w') as f:
            f.write("# rocprofv3 --kernel-trace --stats: python scratch/bvh_prof.py %d  (1 + 6 full builds, then 6 refits of the same topology; MI355X)\n" % P_)
            f.write("name,calls,avg_us\n")
            for r in rows[:20]:
                f.write("%s,%s,%.1f\n" % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3))
for n in ('envgs', 'envgs_reference_caller', 'envgs_twin_caller', 'envgs_f16', 'raster', 'env700k', 'caps', 'config5'):
    src = os.path.join(ROOT, 'gpurun_out', 'bench_%s_final.json' % n)
    if os.path.exists(src) and os.path.getsize(src) > 10:
        open(os.path.join(ROOT, 'profiles', '%s_bench_%s.json' % (TAG, n)), 'w').write(open(src).read())
