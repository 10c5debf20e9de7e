"""Probe: wall time of GPModel.optimize_theta (60 Laplace evidences) and of one evidence at the C2 and C3 shapes."""
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
    np.random.seed(0)
    gp.evidence(list(gp.theta), None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): gp.evidence(list(gp.theta), None)
    torch.cuda.synchronize(); one = (time.perf_counter() - t0) / 5 * 1e3
    th0 = list(gp.theta)
    gp.optimize_theta(); torch.cuda.synchronize()
    gp.theta = th0
    t0 = time.perf_counter(); gp.optimize_theta(); torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
    print(f"{name}: one evidence {one:.2f} ms; optimize_theta {tot:.1f} ms -> theta {np.round(gp.theta, 4)}")
