"""Probe: twenty device-resident mean searches at the C2 shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
class G:
    def __call__(self, n): return load_golden(n)
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
g, gp, st = _fitted(G(), name)
eng, post, pool = gp.eng, gp._mean_post(), gp._candidate_pool()
work = pool.clone()
for _ in range(20):
    eng.shift_points(pool, np.random.uniform(0, 1, gp.D), out=work)
    eng.mean_search(post, work, K=32, sep=5e-2, iters=100, tol=1e-9, sync=False)
torch.cuda.synchronize()
