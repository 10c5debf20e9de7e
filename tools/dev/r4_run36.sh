cd $GRAFT_REPO_ROOT
python -m pytest "tests/test_gpu_incremental.py::test_incremental_replay_matches_cold_refits[whitened]" -x -q -m gpu -s 2>&1 | grep -E "^E |whitened:" | head -12
