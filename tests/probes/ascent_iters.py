"""Probe: iteration counts of ppbo_mean_ascent from 64 uniform starts (C2 and C3 shapes, cap 100 / 400)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
class G:
    def __call__(self, n): return load_golden(n)
for name in ("c2", "c3"):
    g, gp, st = _fitted(G(), name)
    post = gp._mean_post()
    starts = np.random.default_rng(0).random((64, gp.D))
    for cap in (100, 400):
        xs, mus, its = gp.eng.mean_ascent(post, starts, iters=cap, tol=1e-9)
        its = its.cpu().numpy(); mus = mus.cpu().numpy()
        print(f"{name} cap {cap}: iterations min {its.min()} median {int(np.median(its))} max {its.max()}, at cap {(its >= cap).sum()} of 64; best mu {mus.max():.9f}")
    _, g1 = gp.eng.mean_grad(post, xs)
    g1 = g1.cpu().numpy(); x = xs.cpu().numpy()
    pg = np.where(((x <= 0) & (g1 < 0)) | ((x >= 1) & (g1 > 0)), 0.0, g1)
    print("   projected gradient norms after cap 400: max", np.abs(pg).max(axis=1).max(), "median", np.median(np.abs(pg).max(axis=1)))
