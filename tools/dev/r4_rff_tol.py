"""Hsampler.return_xstar's device search (ppbo_rff_search) against the ascent's stopping tolerance: time and the value reached."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
rng = np.random.default_rng(0)
for D, F, ell in ((20, 4096, 0.3), (6, 1000, 0.26)):
    W = rng.standard_normal((F, D)) / ell
    b = rng.uniform(0, 2 * np.pi, F)
    om = rng.standard_normal(F)
    cand = eng.dev(rng.random((65536, D)))
    for tol in (1e-10, 1e-8, 1e-6, 1e-5):
        for iters in (200, 100):
            x, v = eng.rff_search(cand, W, b, 0.5, om, tol=tol, iters=iters)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                x, v = eng.rff_search(cand, W, b, 0.5, om, tol=tol, iters=iters)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
            print(f"D={D} F={F} tol={tol:g} iters={iters}: {dt:.3f} ms, best value {np.max(v):.12f}, found {len(v)}")
