"""Turn the rocprofv3 outputs merged under gpurun_out/ into the small tracked summaries in profiles/ (run after a profiling gpurun call)."""
import collections, csv, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def short(n):
    m = re.search(r'radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|init_lookback_scan_state|radix_sort_block_sort|radix_sort_merge', n)
    if m: return 'rocprim::' + m.group(0)
    n = re.sub(r'\(.*', '', n.replace('void ', '')).replace('at::native::', 'at::')
    return n[:90].replace(',', ';')


def stats(tag, steps, desc):
    import glob as _g
    cand = _g.glob(os.path.join(ROOT, 'gpurun_out', 'p_%s' % tag, '**', '*kernel_stats.csv'), recursive=True)
    if not cand: return
    path = cand[0]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    with open(os.path.join(ROOT, 'profiles', '%s_%s_kernel_stats.csv' % (TAG, tag)), 'w') as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv (MI355X)  -- %s\n" % desc)
        f.write("# %d steps incl. warm-up; sum of all kernel time = %.3f ms per step\n" % (steps, tot / steps / 1e6))
        f.write("name,calls,total_ns,avg_ns,pct,ms_per_step\n")
        for r in rows[:32]:
            f.write("%s,%s,%s,%.0f,%s,%.4f\n" % (short(r['Name']), r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['Percentage'], float(r['TotalDurationNs']) / steps / 1e6))


NAMES = {'envgs::composite_fwd': 'composite_fwd', 'envgs::composite_bwd': 'composite_bwd', 'envgs::project_surfels': 'project_surfels',
         'envgs::project_surfels_bwd': 'project_surfels_bwd', 'envgs::bin_pass': 'bin_tile_pairs', 'envgs::sort_tile_lists': 'sort_tile_lists',
         'envgs::collect_hits_coop': 'trace.collect_hits', 'envgs::sort_composite_fwd<4, false, true>': 'trace.sort_composite_fwd',
         'envgs::sort_composite_fwd<4, false, false>': 'trace.sort_composite_fwd(per-lane SH gathers)',
         'envgs::register_hits': 'trace.register_hits', 'envgs::batch_surfel_bwd': 'trace.batch_surfel_bwd',
         'envgs::batch_surfel_bwd<true, false>': 'trace.batch_surfel_bwd<colour only>', 'envgs::batch_surfel_bwd<false, true>': 'trace.batch_surfel_bwd<generic, others>',
         'envgs::batch_surfel_bwd<false, false>': 'trace.batch_surfel_bwd<generic>',
         'envgs::reduce_surfel_records': 'trace.reduce_surfel_records'}


def pmc(key):
    """profiles/<TAG>_pmc_<key>.json from gpurun_out/pmc_<key>_*/**/counter_collection.csv (raw rocprofv3 rows -- one per launch and counter -- or
    the per-(kernel, counter) averages profiles/shrink_counters.py leaves behind)."""
    import glob
    ctr = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum over launches, launches]
    for path in glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_%s_*' % key, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            kn = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))
            n = int(r['Launches']) if 'Launches' in r else 1
            c = ctr[kn][r['Counter_Name']]
            c[0] += float(r['Counter_Value']) * n; c[1] += n
    if not ctr:
        return
    out = {"_how": "rocprofv3 --pmc <one counter group> --kernel-trace, one pass per group (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* | SQ activity | TCP; no other trace domain) over "
                   "`python bench.py <workload args> --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-reference-caller` on MI355X (profiles/collect_profiles.sh " + key + "); per-launch averages. "
                   "Units per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16 B/lane reads, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. "
                   "Calibrated in round 3 (profiles/r03_fetch_calibration.txt): FETCH_SIZE = read requests x 64 B -- a coalesced stream issues 128 B requests (hence the x2), a scattered 16 / 32 B gather "
                   "ONE request per lane-access, so for the gather-dominated tracer kernels hbm_bytes over-states the reads by up to 2x; fetch_requests = FETCH_SIZE*1024/64 is exact for every kernel. "
                   "A kernel name without template arguments pools all its instantiations (launch-weighted).",
           "workload": key, "kernels": {}}
    for k, v in NAMES.items():
        agg = collections.defaultdict(lambda: [0.0, 0])
        for name, cs in ctr.items():
            if name == k or ('<' not in k and name.startswith(k + '<')):
                for c, (tot, n) in cs.items():
                    agg[c][0] += tot; agg[c][1] += n
        if not agg:
            continue
        e = {c: round(tot / max(n, 1), 1) for c, (tot, n) in sorted(agg.items())}
        fv, wv = e.pop('FETCH_SIZE', None), e.pop('WRITE_SIZE', None)
        row = {}
        if fv is not None:
            row.update({"FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": wv or 0.0, "hbm_bytes": int((2 * fv + (wv or 0.0)) * 1024), "fetch_requests": int(fv * 1024 / 64)})
        row.update(e)
        row["launches_sampled"] = max(n for _, n in agg.values())
        out["kernels"][v] = row
    out["_sq"] = ("SQ_INSTS_* count wave-level instructions, SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles, SQ_BUSY_CU_CYCLES counts cycles summed over CUs "
                  "(MI355X_MICROARCH.md, rocprofv3 PMC slots)")
    json.dump(out, open(os.path.join(ROOT, 'profiles', '%s_pmc_%s.json' % (TAG, key)), 'w'), indent=1)


DESC = {'envgs': (64, 'full EnvGS step (configs[2]): 300k base surfels ch05 raster + 163840 env surfels LBVH trace, 800x800; python bench.py --steps 20 --warmup 4 --repeats 3 (the tracer forward runs as two batch segments on two streams: 2 launches per step, overlapping, so the kernel times sum to more than the step)'),
        'raster': (94, 'raster only (configs[1]): 300k surfels, SH deg 3 in-kernel, 800x800; python bench.py --workload raster --steps 30 --warmup 4 --repeats 3'),
        'config5': (22, 'configs[4]: 1200x1600, -ch07, two bounces, fp16 storage; python bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --steps 6 --warmup 4 --repeats 3'),
        'env700k': (64, 'env set at its 700 000 cap; python bench.py --env-gaussians 700000 --steps 20 --warmup 4 --repeats 3'),
        'base_trace_d0': (64, 'camera rays traced over the 300k base set, all outputs differentiated, bounce-free; python bench.py --workload base_trace --trace-depth 0 --steps 20 --warmup 4 --repeats 3'),
        'base_trace_d2': (64, 'camera rays traced over the 300k base set, all outputs differentiated, two bounces; python bench.py --workload base_trace --trace-depth 2 --steps 20 --warmup 4 --repeats 3')}
for key_, (steps_, desc_) in DESC.items():
    stats(key_, steps_, desc_ + ' --no-cpu-baseline --no-render --no-reference-caller')
    pmc(key_)
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
for n in ('envgs', 'envgs_reference_caller', 'envgs_twin_caller', 'envgs_stream_ordered', 'envgs_f16', 'raster', 'env700k', 'caps', 'config5', 'base_trace_d0', 'base_trace_d2'):
    src = os.path.join(ROOT, 'gpurun_out', 'bench_%s_final.json' % n)
    if os.path.exists(src) and os.path.getsize(src) > 10:
        open(os.path.join(ROOT, 'profiles', '%s_bench_%s.json' % (TAG, n)), 'w').write(open(src).read())
