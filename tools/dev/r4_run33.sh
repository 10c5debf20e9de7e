cd $GRAFT_REPO_ROOT
for lds in 2 10000 20000 26000 30000 34000 40000 50000; do for gf in 8 12; do echo -n "extra_lds=$lds gf_from=$gf: "; PPBO_FIT_GF_FROM=$gf PPBO_FIT_OVERLAP=$lds python tools/fit_only.py c3 z 2>&1 | tail -1 | cut -c1-24; done; done
echo -n "baseline: "; PPBO_FIT_OVERLAP=0 python tools/fit_only.py c3 z 2>&1 | tail -1 | cut -c1-24
