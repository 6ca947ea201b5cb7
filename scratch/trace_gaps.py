"""GPU idle gaps of a step from a rocprofv3 --kernel-trace CSV: python scratch/trace_gaps.py gpurun_out/kt2/kt_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ends = [e for s, e, n in ev if 'fused_adam_multi' in n]
for si in range(len(ends) - 4, len(ends)):
    t0, t1 = ends[si - 1], ends[si]
    ks = [(s, e, n) for s, e, n in ev if s >= t0 and e <= t1]
    ce, idle, big, prev = t0, 0, [], None
    for s, e, n in ks:
        if s > ce:
            idle += s - ce
            if s - ce > 15000: big.append("%.0f us after [%s] before [%s]" % ((s - ce) / 1e3, (prev or '')[:40], n[:40]))
        if e > ce: ce, prev = e, n
    print("step %.3f ms, idle %.3f ms, %d kernels" % ((t1 - t0) / 1e6, idle / 1e6, len(ks)))
    for b in big: print("   ", b)
