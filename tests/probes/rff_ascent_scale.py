"""Probe: ppbo_rff_search time against the iteration cap (C3 shape: F = 4096, D = 20, 65536 candidates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
rng = np.random.default_rng(3)
M, D, F = 65536, 20, 4096
Xc = eng.dev(rng.random((M, D))); W = eng.dev(rng.standard_normal((F, D)) / 0.3); b = eng.dev(rng.uniform(0, 2 * np.pi, F))
om = eng.dev(rng.standard_normal(F))
for iters in (0, 10, 50, 100, 200):
    for K in (32, 8):
        eng.rff_search(Xc, W, b, 0.5, om, K=K, iters=iters)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): xs, vals = eng.rff_search(Xc, W, b, 0.5, om, K=K, iters=iters)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"iters {iters:3d} K {K:2d}: {dt:.3f} ms   best {vals.max():.9f}")
