#!/usr/bin/env python3
"""PPBO on Hartmann6 with the drop-in classes (the reference's run_ppbo_loop, ppbo_numerical_main.py:57-127,
with its hartmann6d setup :174-183): D initial coordinate queries, then PCD / EI-EXT-FAST queries answered by a
simulated user who picks the best point on the projective line.

    python examples/ppbo_hartmann6.py --queries 30 --strategy PCD
    python examples/ppbo_hartmann6.py --queries 30 --compare      # cold refits vs the incremental fit (f-4)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]])
P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886], [2329, 4135, 8307, 3736, 1004, 9991],
                     [2348, 1451, 3522, 2883, 3047, 6650], [4047, 8828, 8732, 5743, 1091, 381]])
ALPHA = np.array([1.0, 1.2, 3.0, 3.2])


def hartmann6(x):
    x = np.atleast_2d(x)
    inner = np.einsum("kd,nkd->nk", A, (x[:, None, :] - P[None, :, :]) ** 2)
    return -(ALPHA[None, :] * np.exp(-inner)).sum(axis=1)       # minimum -3.322 at (0.2017, 0.15, 0.4769, 0.2753, 0.3117, 0.6573)


def user(xi, x, lo, hi):
    from ppbo_amd.misc import alpha_bounds
    a0, a1 = alpha_bounds(xi, lo, hi)
    al = np.linspace(a0, a1, 2001)
    return float(al[np.argmin(hartmann6(al[:, None] * xi[None, :] + x[None, :]))])


def run(queries=30, strategy="PCD", m=31, seed=0, verbose=False, incremental=False, method="whitened"):
    from ppbo_amd.acquisition import next_query
    from ppbo_amd.gp_model import GPModel
    from ppbo_amd.ppbo_settings import PPBO_settings
    D = 6
    bounds = ((0, 1),) * D
    lo, hi = np.zeros(D), np.ones(D)
    np.random.seed(seed)
    st = PPBO_settings(D=D, bounds=bounds, xi_acquisition_function=strategy, m=m, theta_initial=[0.001, 0.26, 0.1],
                       verbose=False)
    xis = np.eye(D)
    xs = np.random.uniform(lo, hi, (D, D))
    results = np.empty((0, 2 * D + 1))
    gp = None
    hist = []
    t0 = time.time()
    for i in range(D):
        if i == D - 1 and gp is not None:
            gp.turn_initialization_off()
        xi, x = xis[i].copy(), xs[i].copy()
        x[xi != 0] = 0
        a = user(xi, x, lo, hi)
        results = np.vstack([results, np.concatenate([a * xi + x, xi, [a]])])
        if gp is None:
            gp = GPModel(st, incremental=incremental)
            gp.fMAP_method = method
        gp.update_feedback_processing_object(results)
        gp.update_data()
        gp.update_model()
    gp.turn_initialization_off()
    for i in range(queries):
        if i + 1 == queries:
            gp.set_last_iteration()
        xi, x = next_query(st, gp, unscale=True)
        a = user(xi, x, lo, hi)
        results = np.vstack([results, np.concatenate([a * xi + x, xi, [a]])])
        gp.update_feedback_processing_object(results)
        gp.mustar_previous_iteration = gp.mustar
        gp.update_data()
        n_log = len(gp.fit_log)
        t_q = time.time()
        gp.update_model()
        t_q = time.time() - t_q
        fx = float(hartmann6(gp.FP.unscale(gp.xstar))[0])
        trials = gp.fit_log[n_log:]
        hist.append(dict(fx=fx, N=gp.N, update_model_s=t_q, fit_s=sum(t["seconds"] for t in trials),
                         iterations=sum(t["iterations"] for t in trials), n_cholesky=sum(t["n_cholesky"] for t in trials),
                         evals=sum(t.get("lbfgs_evals", 0) for t in trials), trials=len(trials)))
        if verbose:
            h = hist[-1]
            print(f"query {i + 1:3d}  N={gp.N:5d}  f(x*)={fx:+.4f}  fit: {h['trials']} start(s), {h['evals']:4d} L-BFGS evaluations, "
                  f"{h['iterations']:4d} TR iterations, {h['n_cholesky']:5d} factorizations, {h['fit_s'] * 1e3:8.1f} ms, "
                  f"update_model {h['update_model_s'] * 1e3:7.1f} ms   elapsed {time.time() - t0:6.1f}s")
    return gp, hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=30)
    ap.add_argument("--strategy", default="PCD")
    ap.add_argument("--m", type=int, default=31)
    ap.add_argument("--incremental", action="store_true", help="bordered Sigma^-1 update + warm-started f_MAP (f-4)")
    ap.add_argument("--compare", action="store_true", help="run cold and incremental back to back and summarise")
    args = ap.parse_args()
    # (incremental, f_MAP method): the trust region alone is what rounds 1-2 ran
    modes = ([(False, "trust-region"), (False, "whitened"), (True, "whitened")] if args.compare
             else [(args.incremental, "whitened")])
    summary = {}
    for inc, method in modes:
        tag = f"{'incremental' if inc else 'cold (reference semantics: prior draw per update)'}, f_MAP by {method}"
        print(f"---- {tag} ----")
        gp, hist = run(args.queries, args.strategy, args.m, verbose=True, incremental=inc, method=method)
        body = hist[:-1] if len(hist) > 1 else hist            # the last query runs the reference's 10 random restarts
        summary[tag] = dict(best=min(h["fx"] for h in hist), fit_ms=1e3 * np.mean([h["fit_s"] for h in body]),
                            upd_ms=1e3 * np.mean([h["update_model_s"] for h in body]),
                            chol=np.mean([h["n_cholesky"] for h in body]), iters=np.mean([h["iterations"] for h in body]),
                            evals=np.mean([h["evals"] for h in body]))
        print("best f(x*) reached:", summary[tag]["best"], "(global minimum -3.322)")
    for tag, s in summary.items():
        print(f"{tag}: mean per query (last excluded) fit {s['fit_ms']:8.2f} ms (update_model {s['upd_ms']:7.2f} ms), "
              f"{s['evals']:6.1f} L-BFGS evaluations, {s['iters']:6.1f} TR iterations, {s['chol']:6.1f} factorizations; "
              f"best f(x*) {s['best']:+.4f}")
