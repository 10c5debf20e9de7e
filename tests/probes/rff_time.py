"""Probe: wall time of the Hsampler cycle (update_phi_X, update_omega_MAP, update_covariancematrix, sample_xstar) at
the C2 (F = 1000) and C3 (F = 4096) shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
from ppbo_amd.random_fourier_sampler import Hsampler
class G:
    def __call__(self, n): return load_golden(n)
def T(fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r
for name, F in (("c2", 1000), ("c3", 4096)):
    g, gp, st = _fitted(G(), name)
    np.random.seed(0)
    gp.xstar, gp.mustar, gp.xstars_local = gp.mu_star()
    hs = Hsampler(gp, F)
    hs.generate_basis()
    t_phi, _ = T(hs.update_phi_X)
    np.random.seed(1)
    t_map, _ = T(hs.update_omega_MAP)
    t_cov, _ = T(hs.update_covariancematrix)
    t_smp, _ = T(hs.sample_xstar)
    print(f"{name} F={F}: update_phi_X {t_phi:.2f} ms, update_omega_MAP {t_map:.2f} ms, update_covariancematrix {t_cov:.2f} ms, sample_xstar {t_smp:.2f} ms")
    print("   omega_MAP stats", hs.omega_MAP_stats)
