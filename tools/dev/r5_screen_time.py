"""Mean-only scoring of 65536 + N + 1 candidates (what a mu_star trial screens) with K* in fp64 and in fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_MEAN
eng = get_engine(0)
for cfg in ("c3", "c2"):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
    X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
    r = eng.gp_fit(X, th, kern, m, g["f_init"])
    post = r["post"]
    D = X.shape[1]
    Xc = eng.dev(np.random.default_rng(1).random((65536 + X.shape[0] + 1, D)))
    out = {}
    for f32 in (False, True):
        fn = lambda: eng.predict(post, Xc, score=SCORE_MEAN, want_var=False, want_best=False, kstar_fp32=f32)
        for _ in range(3): o = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): o = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        out[f32] = (dt, o["mu"].cpu().numpy())
    err = np.abs(out[True][1] - out[False][1]).max() / np.abs(out[False][1]).max()
    top64 = lambda v: set(np.argsort(-v)[:64].tolist())
    print(f"{cfg}: fp64 {out[False][0]*1e3:.3f} ms  fp32-K* {out[True][0]*1e3:.3f} ms  max rel err {err:.2e}  top-64 overlap {len(top64(out[True][1]) & top64(out[False][1]))}/64")
