OUT=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $OUT/tests_full.log 2>&1
tail -3 $OUT/tests_full.log
python __graft_entry__.py --smoke 2>&1 | tail -1
bash tools/dev/r4_final_profiles.sh
