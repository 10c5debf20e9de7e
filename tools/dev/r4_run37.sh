cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-secondary --no-cpu-baseline --no-precision-report --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['gp_fit_ms'])"; done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
