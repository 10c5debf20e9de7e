set -x
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triangular_inverse or potrf" --durations=5 2>&1 | tail -30 > gpurun_out/r4f/tests_inv.log
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 > gpurun_out/r4f/tests.log
python tools/fit_only.py c3 > gpurun_out/r4f/fit_fused.txt 2>&1
python tools/linalg_bench.py > gpurun_out/r4f/linalg.txt 2>&1
PPBO_POTRF_SHADOW=0 python tools/linalg_bench.py > gpurun_out/r4f/linalg_noshadow.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4f/fitprof 260 > gpurun_out/r4f/fit_trace.txt
rm -rf gpurun_out/r4f/fitprof
