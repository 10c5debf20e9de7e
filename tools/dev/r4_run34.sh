cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gp_fit.py tests/test_gpu_dropin.py tests/test_gpu_whitened.py tests/test_gpu_concurrent.py tests/test_gpu_multistart.py tests/test_gpu_incremental.py tests/test_gpu_parity.py tests/test_gpu_c5.py tests/test_gpu_golden_r2.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -12
for c in c2 c3 c4; do python tools/fit_only.py $c z 2>&1 | tail -1 | cut -c1-200; done
for c in c2 c3 c4; do python tools/fit_only.py $c 2>&1 | tail -1 | cut -c1-200; done
