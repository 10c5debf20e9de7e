OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_whitened.py tests/test_gpu_gp_fit.py tests/test_gpu_multistart.py tests/test_gpu_incremental.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for c in c2 c3 c4; do python tools/fit_only.py $c 2>&1 | tail -2; done | tee $OUT/fit.txt
