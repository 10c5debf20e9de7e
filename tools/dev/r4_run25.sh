cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
for v in "2048 3" "1024 3" "512 3" "256 3" "128 3" "512 2"; do
set -- $v
export PPBO_SYRK_SPLIT=$1 PPBO_SYRK_CFG=$2
rocprofv3 --kernel-trace --output-format csv -d $OUT/sy -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > /dev/null 2>&1
echo "== Ks=$1 cfg=$2" >> $OUT/sy_all.txt
python3 $GRAFT_REPO_ROOT/tools/dev/trace_summary.py $OUT/sy 400 | grep -B1 -A0 "mirror" | tail -2 >> $OUT/sy_all.txt
rm -rf $OUT/sy
done
cat $OUT/sy_all.txt
