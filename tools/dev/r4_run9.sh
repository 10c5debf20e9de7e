mkdir -p gpurun_out/r4i
python tools/fit_only.py c3 > gpurun_out/r4i/fit_lowprio.txt 2>&1
PPBO_SIDE_LOW_PRIORITY=0 python tools/fit_only.py c3 > gpurun_out/r4i/fit_plain.txt 2>&1
python tools/fit_only.py c2 > gpurun_out/r4i/fit_c2.txt 2>&1
python tools/fit_only.py c4 > gpurun_out/r4i/fit_c4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4i/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4i/fitprof 260 > gpurun_out/r4i/fit_trace.txt
rm -rf gpurun_out/r4i/fitprof
