"""Probe: optimize_theta wall time against the number of concurrent evidence workers (C2 and C3 shapes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
class G:
    def __call__(self, n): return load_golden(n)
for name in ("c2", "c3"):
    g, gp, st = _fitted(G(), name)
    gp.verbose = False
    th0 = list(gp.theta)
    for W in (2, 4, 8, 16, 32):
        gp.theta = list(th0); np.random.seed(0)
        gp.optimize_theta(workers=W); torch.cuda.synchronize()
        gp.theta = list(th0); np.random.seed(0)
        t0 = time.perf_counter(); gp.optimize_theta(workers=W); torch.cuda.synchronize()
        print(f"{name}: workers {W:2d}: optimize_theta {(time.perf_counter() - t0) * 1e3:.1f} ms -> theta {np.round(gp.theta, 4)}", flush=True)
