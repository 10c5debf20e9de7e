cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
for ch in 1 0; do PPBO_QF_CHAIN=$ch python bench.py --no-secondary --no-cpu-baseline --no-precision-report --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chain=$ch', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done | tee gpurun_out/r4q/chain.txt
