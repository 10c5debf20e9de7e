set -x
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_gp_fit.py tests/test_gpu_whitened.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r4e/tests.log
python tools/fit_only.py c3 > gpurun_out/r4e/fit_fused.txt 2>&1
python tools/fit_only.py c3 calls > gpurun_out/r4e/fit_calls.txt 2>&1
python tools/fit_only.py c2 > gpurun_out/r4e/fit_fused_c2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4e/fitprof 250 > gpurun_out/r4e/fit_trace.txt
rm -rf gpurun_out/r4e/fitprof
