cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gp_fit.py tests/test_gpu_dropin.py tests/test_gpu_whitened.py tests/test_gpu_concurrent.py tests/test_gpu_multistart.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
