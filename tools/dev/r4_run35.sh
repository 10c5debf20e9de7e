cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
rocprofv3 --kernel-trace --output-format csv -d $OUT/ov -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 z > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/trace_summary.py $OUT/ov 230 > $OUT/overlap_trace.txt
rm -rf $OUT/ov
