"""A few sharded-search steps at C3 for M in argv (default 8192 65536): target for rocprofv3 --kernel-trace (where do
the ~90 us of fixed cost per step go?).  Also prints host wall per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_POINTWISE_EI
from ppbo_amd.dist import ShardedSearch
eng = get_engine(0)
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "c3.npz")))
X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
S = eng.gram(X, th, kern)
Sinv, L = eng.pd_inverse_chol(S)
f, _ = eng.fit_fmap(Sinv, g["f_init"], m, th[0], L=L)
post = eng.posterior(X, th, kern, Sinv, f, m)
mustar = float(np.max(g["mu"]))
for M in [int(a) for a in sys.argv[1:]] or [8192, 65536]:
    s = ShardedSearch(eng, post, np.random.default_rng(1).random((M, X.shape[1])), 0, SCORE_POINTWISE_EI, mustar, collective="capi")
    for _ in range(5):
        s.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        s.step()
    torch.cuda.synchronize()
    print(f"M {M}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step (host wall)")
