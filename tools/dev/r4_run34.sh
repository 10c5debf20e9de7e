cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gp_fit.py tests/test_gpu_parity.py tests/test_gpu_compat.py tests/test_gpu_whitened.py tests/test_gpu_incremental.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6
for c in c3 c3 c2; do python tools/fit_only.py $c z 2>&1 | tail -1 | cut -c1-60; done
python tools/fit_only.py c3 tr 2>&1 | tail -1 | cut -c1-100
