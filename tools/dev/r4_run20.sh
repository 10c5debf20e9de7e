OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_c5.py tests/test_gpu_compat.py tests/test_gpu_dropin.py tests/test_gpu_golden_r2.py tests/test_gpu_mean_search.py -x -q -m gpu > $OUT/tests19.log 2>&1
tail -5 $OUT/tests19.log
for c in c2 c3; do python tools/dev/r4_hs_profile.py $c 2>&1 | grep -E "hsampler cycle|rff_omega_map|return_xstar"; done | tee $OUT/hs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/hsprof -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_hs_profile.py c2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py $OUT/hsprof 0 > $OUT/hs_trace_c2.txt
rm -rf $OUT/hsprof
head -30 $OUT/hs_trace_c2.txt
