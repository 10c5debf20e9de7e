"""Y = G K* of the line acquisition alone (library event brackets), for PPBO_LINE_Y_CHUNK experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "c3.npz")))
X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
post = eng.gp_fit(X, th, kern, m, g["f_init"])["post"]
D = X.shape[1]
B, G, S = 512, 70, 150
rng = np.random.default_rng(6)
xis = np.eye(D)[np.arange(B) % D]
xs = rng.random((B, D)); xs[np.arange(B), np.arange(B) % D] = 0.0
z = eng.dev(rng.standard_normal((S, G)))
xd, sd, ad = eng.dev(xis), eng.dev(xs), eng.dev(np.linspace(0.005, 0.995, G))
mustar = float(np.max(g["mu"]))
out = None
for rep in range(8):
    if rep == 3: eng.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.line_acq_xi(post, xd, sd, ad, z, mustar, jitter=1e-10 * float(th[2]) ** 2)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
ms = {k: eng.profile_read(k) for k in ("line_kstar", "line_y", "line_cov", "line_mc")}
print(f"PPBO_LINE_Y_CHUNK={os.environ.get('PPBO_LINE_Y_CHUNK', 'auto')}: call {t:.3f} ms | " +
      " ".join(f"{k} {v[0] / max(v[1], 1):.3f}" for k, v in ms.items()) + f" | ei sum {float(out[0].sum()):.12e} vm sum {float(out[1].sum()):.12e}")
