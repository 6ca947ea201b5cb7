cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_envgs_step_parity.py tests/test_fp16_storage.py tests/test_train_convergence.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
