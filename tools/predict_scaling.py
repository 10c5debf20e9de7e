"""Predicted 1/2/4/8-GPU numbers for bench.py, from ONE GPU: every rank's leg is run here (the model is replicated and
the candidates are independent, so a rank's step time does not depend on the others), plus the measured latency of
the one collective at world = 1 and the published xGMI all-gather latency range for 16-byte records.

    python tools/predict_scaling.py > profiles/r03_scaling_prediction.txt

C4 (BASELINE config 4): 262144 candidates split over G ranks (strong scaling).  C3: 65536 candidates per rank (weak).
The first real SCALE run has this table to be compared with."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_POINTWISE_EI
from ppbo_amd.dist import shard_bounds

eng = get_engine(0)


def fitted(cfg):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"{cfg}.npz")))
    X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
    S = eng.gram(X, th, kern)
    Sinv, L = eng.pd_inverse_chol(S)
    f, _ = eng.fit_fmap(Sinv, g["f_init"], m, th[0], L=L)
    return g, eng.posterior(X, th, kern, Sinv, f, m)


def step_ms(post, Xc, mustar, steps=20):
    def step():
        return eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=mustar, want_mu=False, want_var=False,
                           want_score=False, want_best=True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


# the collective at world = 1 (host round trip of ppbo_argmax_allgather without the RCCL hop is not available
# stand-alone; its device part is the one-wavefront combine + one 16-byte read-back, measured here)
rec = eng.dev(np.array([[1.0, 5.0]]))
for _ in range(10):
    eng.argmax_combine(rec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.argmax_combine(rec)
combine_us = (time.perf_counter() - t0) / 200 * 1e6
# a 16-byte all-gather over xGMI is latency-bound: RCCL's LL protocol needs ~15-30 us at 8 ranks on one node
ag_lo, ag_hi = 15.0, 30.0
print(f"# device-side combine + 16-byte read-back, measured: {combine_us:.1f} us per step; RCCL all-gather of 16 B/rank "
      f"over xGMI assumed {ag_lo:.0f}-{ag_hi:.0f} us (latency-bound, not measurable on a 1-GPU box)")

print("\n# C4 (N=1024, D=10), 262144 candidates in total, STRONG scaling: rank legs measured one after the other on one GPU")
print("# G   M/rank   step ms (slowest rank)   predicted evals/s (all ranks)   efficiency vs G=1")
g, post = fitted("c4")
D = g["X"].shape[1]
mustar = float(np.max(g["mu"]))
base = None
for G in (1, 2, 4, 8):
    worst = 0.0
    for r in range(G):
        lo, hi = shard_bounds(262144, r, G)
        Xc = eng.dev(np.random.default_rng(1 + r).random((hi - lo, D)))
        worst = max(worst, step_ms(post, Xc, mustar, 10))
        del Xc
    t_lo, t_hi = worst + (combine_us + (ag_lo if G > 1 else 0.0)) * 1e-3, worst + (combine_us + (ag_hi if G > 1 else 0.0)) * 1e-3
    v_lo, v_hi = 262144 / (t_hi * 1e-3), 262144 / (t_lo * 1e-3)
    base = base or v_hi
    print(f"  {G}   {262144 // G:6d}   {worst:8.3f}                 {v_lo:.3e} - {v_hi:.3e}            {v_lo / base / G:.2f} - {v_hi / base / G:.2f}")

print("\n# C3 (N=2048, D=20), 65536 candidates PER RANK, WEAK scaling (bench.py's default): every rank runs the same leg")
g, post = fitted("c3")
D = g["X"].shape[1]
mustar = float(np.max(g["mu"]))
Xc = eng.dev(np.random.default_rng(1).random((65536, D)))
t1 = step_ms(post, Xc, mustar, 20)
print("# G   step ms   predicted evals/s (all ranks)   efficiency vs G=1")
for G in (1, 2, 4, 8):
    t_lo, t_hi = t1 + (combine_us + (ag_lo if G > 1 else 0.0)) * 1e-3, t1 + (combine_us + (ag_hi if G > 1 else 0.0)) * 1e-3
    print(f"  {G}   {t1:7.3f}   {G * 65536 / (t_hi * 1e-3):.3e} - {G * 65536 / (t_lo * 1e-3):.3e}      {t1 / t_hi:.3f} - {t1 / t_lo:.3f}")
