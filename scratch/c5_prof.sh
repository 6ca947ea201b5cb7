#!/bin/bash
# kernel trace of the configs[4] workload (1200x1600, -ch07, two bounces, fp16 storage): per-kernel totals of a few steps + the idle gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_c5
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_c5 -o c5 -- python $R/bench.py --height 1200 --width 1600 --trace-depth 2 --channels 7 --feature-dtype f16 --no-cpu-baseline --no-render --no-reference-caller --steps 4 --warmup 2 --repeats 1 > $O/c5_bench.json 2>/dev/null
cd $R
python scratch/step_inventory.py $(find $O/p_c5 -name "*kernel_trace.csv" | head -1) 40 > $O/c5_inventory.txt 2>&1
python scratch/trace_gaps.py $(find $O/p_c5 -name "*kernel_trace.csv" | head -1) >> $O/c5_inventory.txt 2>&1
cp $(find $O/p_c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv
rm -rf $O/p_c5
head -c 400 $O/c5_bench.json
