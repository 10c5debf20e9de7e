"""potrf(N) and the C3 fit, many repetitions -- target of an A/B between two builds of the library (tools/dev/r6_potrf_ab.sh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
out = []
for N in (512, 2048, 4096):
    rng = np.random.default_rng(N)
    Q = rng.standard_normal((N, N))
    A = eng.dev(Q @ Q.T + N * np.eye(N))
    B = A.clone()
    for _ in range(3):
        B.copy_(A); eng.potrf_(B)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        B.copy_(A); torch.cuda.synchronize(); t0 = time.perf_counter(); eng.potrf_(B); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(f"potrf {N}: {np.median(ts) * 1e3:.3f}")
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "c3.npz")))
X = eng.dev(g["X"]); m = int(g["m"]); th = g["theta"]; kern = str(g["kernel"])
z0 = eng.dev(np.random.default_rng(2).standard_normal(X.shape[0]))
ts = []
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
out.append(f"fit c3: {np.median(ts[2:]) * 1e3:.3f}")
print(" | ".join(out))
