"""Probe: ten Hsampler.sample_xstar at C3 (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
from ppbo_amd.random_fourier_sampler import Hsampler
class G:
    def __call__(self, n): return load_golden(n)
g, gp, st = _fitted(G(), "c3")
np.random.seed(0)
gp.xstar, gp.mustar, gp.xstars_local = gp.mu_star()
hs = Hsampler(gp, 4096); hs.generate_basis(); hs.update_phi_X(); hs.update_omega_MAP(); hs.update_covariancematrix()
for _ in range(10): hs.sample_xstar()
torch.cuda.synchronize()
