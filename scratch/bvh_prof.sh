cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_bvh
cd $R
for P in 163840 700000; do
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bvh -o bvh$P -- python scratch/bvh_prof.py $P > /dev/null 2>&1
f=$(ls $O/p_bvh/*/bvh${P}_kernel_stats.csv 2>/dev/null | head -1)
echo "== P=$P"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print("%-60s calls %4s avg_us %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
