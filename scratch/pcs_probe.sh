#!/bin/bash
# exploratory: does rocprofv3 PC sampling work on this box, and what do its files look like?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for m in host_trap stochastic; do
  if [ $m = host_trap ]; then U=time; I=${1:-100}; else U=cycles; I=${2:-1048576}; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $U --pc-sampling-method $m --pc-sampling-interval $I --kernel-trace \
      -d /tmp/pcs_$m --output-format csv -- python $R/bench.py --no-cpu-baseline --no-render --no-reference-caller --steps 6 --warmup 2 > /tmp/pcs_$m.out 2> /tmp/pcs_$m.err
  echo "== $m rc=$?"; tail -5 /tmp/pcs_$m.err
  find /tmp/pcs_$m -type f | head -20
  for f in $(find /tmp/pcs_$m -type f -name "*pc_sampling*"); do echo "-- $f ($(wc -l < $f) lines, $(du -h $f | cut -f1))"; head -5 $f | cut -c1-600; done
done
