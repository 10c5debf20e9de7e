"""One GP fit (Gram, inverse + Cholesky factor, whitened f_MAP search, posterior) a few times -- target for
rocprofv3 --kernel-trace.   python tools/fit_only.py [c3] [tr|calls|z]   ('z' = start from a whitened draw z0, the drop-in's default; 'tr' = the trust region alone, rounds 1-2's fit; 'calls' = the whitened fit call by call)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
whitened = not (len(sys.argv) > 2 and sys.argv[2] == "tr")
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"{name}.npz")))
X = eng.dev(g["X"]); m = int(g["m"]); th = g["theta"]; kern = str(g["kernel"])
f0 = eng.dev(g["f_init"])
fused = whitened and not (len(sys.argv) > 2 and sys.argv[2] == "calls")
zstart = len(sys.argv) > 2 and sys.argv[2] == "z"          # the product's start: z0 ~ N(0, I), f_init = L z0 (GPModel._fit_fused)
z0 = eng.dev(np.random.default_rng(2).standard_normal(X.shape[0]))
for rep in range(4):
    if fused:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True) if zstart else eng.gp_fit(X, th, kern, m, f0)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"rep {rep}: total {(t3 - t0) * 1e3:.2f} ms (ppbo_gp_fit, one call)", r["stats"])
        continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = eng.gram(X, th, kern)
    if whitened:
        Sinv, L = eng.pd_inverse_chol(S)
    else:
        Sinv, L = eng.pd_inverse(S), None
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f, st = eng.fit_fmap(Sinv, f0, m, th[0], L=L)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    post = eng.posterior(X, th, kern, Sinv, f, m)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep {rep}: total {(t3 - t0) * 1e3:.2f} ms = Sigma+inverse {(t1 - t0) * 1e3:.2f} + f_MAP {(t2 - t1) * 1e3:.2f} + posterior {(t3 - t2) * 1e3:.2f}", st)
