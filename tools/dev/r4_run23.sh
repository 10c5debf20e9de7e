cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
python tools/dev/r4_liney.py 2>&1 | tail -1 | tee gpurun_out/r4q/liney3.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_acq_search.py tests/test_gpu_whitened.py tests/test_gpu_gp_fit.py tests/test_gpu_incremental.py tests/test_gpu_c5.py -x -q -m gpu 2>&1 | tail -3
