cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_acq_search.py tests/test_gpu_golden_r2.py -x -q -m gpu -k "line or acq or cov or EI or varmax" 2>&1 | tail -2
python tools/dev/r4_liney.py 2>&1 | tail -1
python tools/dev/r4_liney.py 2>&1 | tail -1
