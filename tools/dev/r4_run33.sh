cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do if [ $p = 1 ]; then export PPBO_SIDE_PRIO=1; else unset PPBO_SIDE_PRIO; fi; echo -n "side low priority=$p: "; python tools/fit_only.py c3 z 2>&1 | tail -1 | cut -c1-24; done
