"""Round 6: 512 lines x 70 points x 150 draws through ppbo_line_acq_xi at the C3 shape: wall time and per-kernel times."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ppbo_amd.engine import Engine  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
e = Engine(0)
X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
N, D = X.shape
r = e.gp_fit(e.dev(X), th, kern, m, e.dev(g["f_init"]), gtol=1e-4)
post = r["post"]
B, G, S = 512, 70, 150
rng = np.random.default_rng(6)
al = np.linspace(0.005, 0.995, G)
xs = rng.random((B, D)); dsel = np.arange(B) % D
z = e.dev(rng.standard_normal((S, G)))
xi_l, x_l = np.eye(D)[dsel], xs.copy(); x_l[np.arange(B), dsel] = 0.0
xi_d, x_d, al_d = e.dev(xi_l), e.dev(x_l), e.dev(al)
mustar = float(np.max(g["mu"]))
fn = lambda: e.line_acq_xi(post, xi_d, x_d, al_d, z, mustar, jitter=1e-10 * float(th[2]) ** 2)
for _ in range(3): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fn()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
e.profile(True)
for _ in range(5): fn()
torch.cuda.synchronize()
lk = {k: e.profile_read(k) for k in ("line_kstar", "line_y", "line_cov", "line_mc")}
print(f"{cfg}: wall {t*1e3:.3f} ms | " + " ".join(f"{k} {v[0]/max(v[1],1):.3f}" for k, v in lk.items()))
ei, vm = fn()
print("checksum", float(ei.sum()), float(vm.sum()))
