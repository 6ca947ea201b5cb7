#!/bin/bash
# refresh the rocprofv3 kernel stats of the default bench command only (the last GPU-seconds of a round)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/p_envgs
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_envgs -o envgs -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 20 --warmup 4 > /dev/null 2>&1
ls $O/p_envgs/* | head -3
