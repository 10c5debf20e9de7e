set -x
mkdir -p gpurun_out/r4b
python tools/dev/r4_step_trace.py 8192 16384 65536 > gpurun_out/r4b/step_wall.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4b/t8192 -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_step_trace.py 8192 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4b/t65536 -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_step_trace.py 65536 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4b/t8192 12 > gpurun_out/r4b/sum8192.txt
python tools/dev/trace_summary.py gpurun_out/r4b/t65536 12 > gpurun_out/r4b/sum65536.txt
rm -rf gpurun_out/r4b/t8192 gpurun_out/r4b/t65536
