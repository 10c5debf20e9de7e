"""Probe: wall time of GPModel.mu_star at the C2 shape for 1 / 3 / 10 trials, and of its pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
class G:
    def __call__(self, n): return load_golden(n)
g, gp, st = _fitted(G(), "c2")
np.random.seed(40)
gp.mu_star(mustar_finding_trials=1); torch.cuda.synchronize()
pol = {"n": 0}
orig = gp._polish
def counted(x):
    pol["n"] += 1
    return orig(x)
gp._polish = counted
for trials in (1, 3, 10):
    for rep in range(3):
        pol["n"] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.mu_star(mustar_finding_trials=trials)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(f"trials {trials}: {dt:.2f} ms total, {dt / trials:.2f} per trial, polishes {pol['n']}")
eng = gp.eng
post = gp._mean_post()
pool = gp._candidate_pool()
work = pool.clone()
for K in (32,):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        eng.shift_points(pool, np.random.uniform(0, 1, gp.D), out=work)
        eng.mean_search(post, work, K=K, sep=5e-2, iters=100, tol=1e-9, sync=False)
    torch.cuda.synchronize(); print(f"10 queued searches: {(time.perf_counter() - t0) * 100:.3f} ms each")
    t0 = time.perf_counter()
    for _ in range(10):
        eng.shift_points(pool, np.random.uniform(0, 1, gp.D), out=work)
        eng.mean_search(post, work, K=K, sep=5e-2, iters=100, tol=1e-9, sync=True)
    torch.cuda.synchronize(); print(f"10 synchronous searches: {(time.perf_counter() - t0) * 100:.3f} ms each")
