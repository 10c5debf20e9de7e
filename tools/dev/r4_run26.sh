cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_whitened.py tests/test_gpu_gp_fit.py tests/test_gpu_multistart.py tests/test_gpu_incremental.py tests/test_gpu_dropin.py tests/test_gpu_c5.py tests/test_gpu_golden_r2.py -x -q -m gpu 2>&1 | tail -2
for c in c2 c3 c4; do python tools/fit_only.py $c 2>&1 | tail -1 | cut -c1-60; done
python tools/fit_whitened.py c3 1e-4 2 2>&1 | grep -E "judgement" | head -3
