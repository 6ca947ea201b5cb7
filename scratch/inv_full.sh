cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_inv
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/p_inv -o inv -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 6 --warmup 3 --repeats 1 > /dev/null 2>&1
cd $R
python scratch/step_inventory.py $(find $O/p_inv -name "*kernel_trace.csv" | head -1) 0 > $O/step_timeline_full.txt 2>&1
rm -rf $O/p_inv
