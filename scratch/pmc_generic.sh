# bash scratch/pmc_generic.sh "<counters>" tag  -> per-kernel averages of an arbitrary counter set (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gpmc_$2
rm -rf $O
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O -o c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-reference-caller > $O.log 2>&1
python - "$O" <<'PY'
import csv, glob, re, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))
        if k.startswith('envgs::') and any(s in k for s in ('collect_hits', 'sort_composite', 'batch_surfel', 'register_hits', 'composite_bwd', 'composite_fwd')):
            acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print('%-40s %-28s n=%d avg=%.5g' % (k[:40], c, len(v), sum(v) / len(v)))
PY
