"""Randomised shapes through the one-launch scoring kernel (PPBO_FUSED=2: both workgroup forms) against the three-launch
form: star sizes 2..97, up to 1024 rows, D 1..24 (and up to 64), SE / RQ, candidate counts off every tile.
   python tests/probes/fused_fuzz.py [cases=80] [seed=0]        (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fused import _engine, synth_post, dense_reference, host  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
e2, e0 = _engine(2), _engine(0)
worst = 0.0
n_fused = 0
for k in range(cases):
    mblk = int(rng.integers(2, 98))
    n_q = int(rng.integers(1, max(2, 1024 // mblk) + 1))
    N = mblk * n_q
    D = int(rng.integers(1, 25)) if k % 5 else int(rng.integers(25, 65))
    kern = "SE_kernel" if rng.random() < 0.6 else "RQ_kernel"
    M = int(rng.integers(1, 3000))
    th = (float(10 ** rng.uniform(-3, 0)), float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.1, 2.0)))
    p2, arrs = synth_post(e2, N, D, mblk - 1, kern, th, seed=k)
    p0, _ = synth_post(e0, N, D, mblk - 1, kern, th, seed=k)
    Xc = rng.random((M, D))
    kind = int(rng.integers(0, 3))
    e2.profile(True)
    o2 = e2.predict(p2, Xc, score=kind, mustar=0.05, want_score=True)
    used = e2.profile_read("fused_score")[1]
    e2.profile(False)
    o0 = e0.predict(p0, Xc, score=kind, mustar=0.05, want_score=True)
    n_fused += used
    err = 0.0
    for key in ("mu", "var", "score"):
        a, b = host(o2[key]), host(o0[key])
        err = max(err, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)))
    mu_ref, var_ref = dense_reference(arrs, Xc, kern, th, mblk - 1)
    err_ref = max(float(np.abs(host(o2["mu"]) - mu_ref).max() / max(np.abs(mu_ref).max(), 1e-300)),
                  float(np.abs(host(o2["var"]) - var_ref).max() / max(np.abs(var_ref).max(), 1e-300)))
    sc = host(o2["score"])
    ok_arg = o2["best_idx"] == int(np.argmax(sc))
    worst = max(worst, err, err_ref)
    flag = "" if (err < 1e-11 and err_ref < 1e-10 and ok_arg) else "   <-- LOOK"
    print(f"{k:3d} N={N:4d} m+1={mblk:2d} D={D:2d} {kern[:2]} M={M:4d} kind={kind} one-launch={used}: vs three-launch {err:.1e}, vs dense {err_ref:.1e}, argmax {ok_arg}{flag}", flush=True)
print(f"{cases} cases, {n_fused} through the one-launch kernel, worst relative difference {worst:.2e}")
