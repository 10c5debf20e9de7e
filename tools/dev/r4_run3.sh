set -x
mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4c/tests.log
python tools/predict_scaling.py > gpurun_out/r4c/scaling.txt 2>gpurun_out/r4c/scaling.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4c/t8192 -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_step_trace.py 8192 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4c/t8192 10 > gpurun_out/r4c/sum8192.txt
rm -rf gpurun_out/r4c/t8192
