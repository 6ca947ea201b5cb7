cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1; tail -2 gpurun_out/gpu_suite.log
