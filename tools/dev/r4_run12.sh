mkdir -p gpurun_out/r4k
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_r2.py -m gpu -q -x -k "line or EI or varmax or acq" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 > gpurun_out/r4k/tests.log
python tools/dev/r4_line_trace.py > gpurun_out/r4k/line.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4k/prof -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_line_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py gpurun_out/r4k/prof 9 > gpurun_out/r4k/line_trace.txt
rm -rf gpurun_out/r4k/prof
