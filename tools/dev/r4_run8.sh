set -x
mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4h/tests.log
python tools/fit_only.py c3 > gpurun_out/r4h/fit_fused.txt 2>&1
PPBO_SIDE_CU_MASK=0 python tools/fit_only.py c3 > gpurun_out/r4h/fit_fused_nomask.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4h/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4h/fitprof 260 > gpurun_out/r4h/fit_trace.txt
rm -rf gpurun_out/r4h/fitprof
