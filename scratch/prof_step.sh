#!/bin/bash
# kernel trace of the default EnvGS workload + the inventory of one step:  bash scratch/prof_step.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/p_step
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_step -o st -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 12 --warmup 4 "$@" > /dev/null 2>&1
python $R/scratch/step_inventory.py $R/gpurun_out/p_step/st_kernel_trace.csv 30 | tail -60
