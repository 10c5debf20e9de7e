cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gp_fit.py tests/test_gpu_whitened.py tests/test_gpu_compat.py tests/test_gpu_incremental.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
for c in c2 c3 c4; do python tools/fit_only.py $c 2>&1 | tail -1 | cut -c1-60; done
