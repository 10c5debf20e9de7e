set -x
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triangular_inverse or potrf" 2>&1 | tail -5 > gpurun_out/r4g/tests_inv.log
python tools/linalg_bench.py > gpurun_out/r4g/linalg.txt 2>&1
python tools/fit_only.py c3 > gpurun_out/r4g/fit_fused.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4g/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4g/fitprof 260 > gpurun_out/r4g/fit_trace.txt
rm -rf gpurun_out/r4g/fitprof
