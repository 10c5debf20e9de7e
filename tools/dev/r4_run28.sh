cd $GRAFT_REPO_ROOT
for sx in 0 1 0 1; do echo "SX=$sx"; PPBO_KSTAR_SX=$sx python tools/kstar_time.py 2>&1 | grep "kstar avg"; done
PPBO_KSTAR_SX=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
