#!/bin/bash
# kernel-time split of the configs[4] step: library kernels vs torch glue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_c5
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_c5 -o c5 -- python $R/bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-reference-caller --no-render --steps 4 --warmup 2 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/p_c5/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 6
cat = collections.Counter(); calls = collections.Counter()
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs']) / steps / 1e6
    k = 'envgs' if 'envgs::' in n else ('torch' if ('at::' in n or 'rocprim' in n or 'hipcub' in n or 'Cijk' in n) else 'copy/fill')
    cat[k] += t; calls[k] += int(r['Calls']) / steps
print({k: (round(v, 2), round(calls[k])) for k, v in cat.items()})
for r in rows[:45]:
    print("%8.3f ms/step %6d  %s" % (float(r['TotalDurationNs']) / steps / 1e6, int(r['Calls']) // steps, r['Name'][:110]))
PY
