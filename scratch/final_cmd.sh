cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_driver_cmd.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r5_driver_cmd.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?"
timeout 300 python bench.py > gpurun_out/bench_envgs_final.json 2> gpurun_out/bench_envgs_final.err; echo "bench rc=$?"
grep -n "passed\|failed" gpurun_out/r5_driver_cmd.log | tail -2
