OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $OUT
python tools/dev/r4_mustar_trace.py c3 > $OUT/mustar.txt 2>&1
python tools/dev/r4_mustar_trace.py c2 >> $OUT/mustar.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/tools/dev/r4_mustar_trace.py c3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py $OUT/prof 60 > $OUT/mustar_trace.txt
rm -rf $OUT/prof
