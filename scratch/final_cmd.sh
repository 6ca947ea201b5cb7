cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_driver_cmd.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r5_driver_cmd.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?"
timeout 400 python bench.py > gpurun_out/bench_envgs_final.json 2> gpurun_out/bench_envgs_final.err; echo "bench rc=$?"
tail -1 gpurun_out/r5_driver_cmd.log; grep -c . gpurun_out/bench_envgs_final.json
