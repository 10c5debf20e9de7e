cd $GRAFT_REPO_ROOT
python tools/fit_whitened.py c3 1e-4 2 2>&1 | grep -E "judgement|whitened\]" | head -12
