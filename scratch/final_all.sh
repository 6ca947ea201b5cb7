#!/bin/bash
# the round's last call: driver command + smoke, then the whole profile collection
cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_driver_cmd.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r5_driver_cmd.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?"
grep -n "passed\|failed" gpurun_out/r5_driver_cmd.log | tail -2
bash profiles/collect_profiles.sh > gpurun_out/collect.log 2>&1; echo "collect rc=$?"
