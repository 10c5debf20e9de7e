"""Probe: where one maximize_EI at the C2 shape spends its time (line_acq calls vs host work)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
import ppbo_amd.acquisition as acq
from test_gpu_golden_r2 import _fitted

class G:
    def __call__(self, n): return load_golden(n)
g, gp, st = _fitted(G(), "c2")
x = load_golden("c2_x")
gp.xstar, gp.mustar = x["xstar"].copy(), float(x["mustar"])
orig = acq._line_scores
log = []
def timed(xis, xs, GP_model, mc, z=None, alphas=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(xis, xs, GP_model, mc, z=z, alphas=alphas)
    torch.cuda.synchronize(); log.append((len(xis), mc, (time.perf_counter() - t0) * 1e3))
    return out
acq._line_scores = timed
for rep in range(3):
    log.clear()
    np.random.seed(rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acq.maximize_EI([0, 1], gp, st)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
    print(f"rep {rep}: total {tot:.2f} ms; line_acq calls (lines, draws, ms):", [(a, b, round(c, 2)) for a, b, c in log], "sum", round(sum(c for _, _, c in log), 2))
# pieces of one call
eng = gp.eng
xis = np.zeros((256, gp.D)); xis[:, 0] = 1.0
xs = np.random.rand(256, gp.D); xs[:, 0] = 0
al = acq._noisy_alphas()
z = np.random.standard_normal((1200, 70))
t0 = time.perf_counter(); grids = al[None, :, None] * xis[:, None, :] + xs[:, None, :]; t1 = time.perf_counter()
zd = eng.dev(z); gd = eng.dev(grids); torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(3):
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ei, vm = eng.line_acq(gp._post, gd, zd, gp.mustar, gp.COVARIANCE_SHRINKAGE, jitter=1e-12)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"grids host {1e3*(t1-t0):.2f} ms, upload {1e3*(t2-t1):.2f} ms, line_acq(256 x 1200, device inputs) {1e3*(t4-t3):.2f} ms")
t0 = time.perf_counter(); np.random.standard_normal((4800, 70)); print(f"host randn 4800x70: {1e3*(time.perf_counter()-t0):.2f} ms")
eng.profile(True)
eng.line_acq(gp._post, gd, zd, gp.mustar, gp.COVARIANCE_SHRINKAGE, jitter=1e-12); torch.cuda.synchronize()
for k in ("line_acq", "predict_cov", "line_mc", "line_prior", "kstar", "gemm"):
    try: print(k, eng.profile_read(k))
    except Exception as e: print(k, "n/a")
