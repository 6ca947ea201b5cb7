cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_inv
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/p_inv -o inv -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 6 --warmup 3 > /dev/null 2>&1
cd $R
f=$(find $O/p_inv -name "*kernel_trace.csv" | head -1)
python scratch/step_inventory.py $f 30 > $O/inventory.txt 2>&1
python scratch/trace_gaps.py $f >> $O/inventory.txt 2>&1
cat $O/inventory.txt
