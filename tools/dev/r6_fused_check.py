"""Round 6: the one-launch scoring kernel (csrc/fused.hip) against the three-launch form on synthetic posterior states.
   python tools/dev/r6_fused_check.py            (on a GPU box)
Prints max differences (mu, var, score) and event-timed step durations of both forms for several shapes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ppbo_amd.engine import Engine, Posterior  # noqa: E402


def synth_post(eng, N, D, m, kernel, theta, seed=0):
    rng = np.random.default_rng(seed)
    mblk = m + 1
    X = rng.random((N, D))
    alpha = rng.standard_normal(N)
    lam_diag = -np.abs(rng.standard_normal(N)) * 0.3
    lam_off = np.abs(rng.standard_normal(N)) * 0.1
    lam_off[::mblk] = 0.0
    G = rng.standard_normal((N, N)) * 0.05
    # block lower triangular with explicit zeros right of the row's star
    r = np.arange(N)
    kend = ((r // mblk) + 1) * mblk
    G[np.arange(N)[None, :] >= kend[:, None]] = 0.0
    return Posterior(kernel, tuple(theta), m, eng.dev(X), eng.dev(alpha), eng.dev(lam_diag), eng.dev(lam_off), eng.dev(G))


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    os.environ["PPBO_FUSED"] = "1"
    ef = Engine(0)
    os.environ["PPBO_FUSED"] = "0"
    e3 = Engine(0)
    shapes = [  # N, D, m, kernel, M
        (64, 2, 31, "SE_kernel", 512), (78, 4, 25, "SE_kernel", 100), (200, 5, 9, "RQ_kernel", 1000),
        (512, 6, 31, "SE_kernel", 16384), (512, 6, 31, "camphor_copper_kernel", 4096), (520, 3, 25, "SE_kernel", 5000),
        (650, 2, 25, "SE_kernel", 16384), (1024, 10, 31, "SE_kernel", 65536), (1014, 7, 25, "RQ_kernel", 3000),
        (1024, 20, 31, "SE_kernel", 16384), (512, 20, 31, "SE_kernel", 65536), (256, 6, 31, "SE_kernel", 16384),
        (1024, 10, 31, "SE_kernel", 262144),
    ]
    for (N, D, m, kern, M) in shapes:
        th = (0.001, 0.26, 0.1) if kern != "RQ_kernel" else (0.3, 0.6, 0.8)
        pf, p3 = synth_post(ef, N, D, m, kern, th), synth_post(e3, N, D, m, kern, th)
        Xc = np.random.default_rng(1).random((M, D))
        xf, x3 = ef.dev(Xc), e3.dev(Xc)
        mustar = 0.1
        of = ef.predict(pf, xf, score=1, mustar=mustar, want_score=True)
        o3 = e3.predict(p3, x3, score=1, mustar=mustar, want_score=True)
        d = {k: float((of[k] - o3[k]).abs().max() / max(float(o3[k].abs().max()), 1e-300)) for k in ("mu", "var", "score")}
        same_best = (of["best_idx"] == o3["best_idx"])
        sf = of["score"].cpu().numpy()
        ok_arg = of["best_idx"] == int(np.argmax(sf))
        reps = 50 if M * N < 3e7 else 10
        tf = timed(lambda: ef.predict(pf, xf, score=1, mustar=mustar, want_mu=False, want_var=False), reps)
        t3 = timed(lambda: e3.predict(p3, x3, score=1, mustar=mustar, want_mu=False, want_var=False), reps)
        print(f"N={N:5d} D={D:2d} m={m:2d} {kern[:3]} M={M:6d}: rel diff mu {d['mu']:.1e} var {d['var']:.1e} score {d['score']:.1e} "
              f"best same {same_best} argmax ok {ok_arg} | fused {tf*1e3:8.1f} us  three-launch {t3*1e3:8.1f} us  ({t3/tf:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
