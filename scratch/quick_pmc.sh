# quick FETCH_SIZE pass over 3 bench steps (GPU box): bash scratch/quick_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qpmc_$1
rm -rf $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render > /dev/null 2>&1
python - "$O" <<'PY'
import csv, glob, re, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))
        if k.startswith('envgs::'): acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print('%-44s %-14s n=%d avg=%.4g KB -> %.3f GB fetched (x2 rule)' % (k[:44], c, len(v), sum(v) / len(v), 2 * sum(v) / len(v) * 1024 / 1e9))
PY
