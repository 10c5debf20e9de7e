"""Randomised shapes through ppbo_predict / ppbo_line_acq against the oracle: star size, number of queries, dimension and
candidate counts drawn at random (padded and unpadded paths of the block-triangular products).  A probe, not a test:
python tests/probes/ragged_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import ppbo_oracle as orc
from ppbo_amd.engine import get_engine
eng = get_engine(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
host = lambda t: t.cpu().numpy()
worst = 0.0
for c in range(cases):
    m = int(rng.integers(1, 48)); n_q = int(rng.integers(1, max(2, 700 // (m + 1)))); D = int(rng.integers(1, 9))
    M = int(rng.choice([1, 3, 70, 127, 128, 129, 500, 2047, 2048, 2049, 3000, 5000]))
    kernel = ("SE_kernel", "RQ_kernel")[int(rng.integers(0, 2))]
    th = [float(10 ** rng.uniform(-1.5, 0)), float(0.3 * np.sqrt(D) * 10 ** rng.uniform(-0.3, 0.3)), float(10 ** rng.uniform(-0.5, 0.3))]
    X = orc.synthetic_design(n_q, D, m=m, seed=c)
    N = X.shape[0]
    S0 = orc.gram(X, th, kernel); Sinv0 = orc.pd_inverse(S0)
    f_init = np.random.default_rng(c).multivariate_normal(np.zeros(N), S0, method="cholesky")
    f0, _ = orc.fit_fmap_trust_exact(f_init, Sinv0, m, th[0], gtol=1e-9)
    try:
        post = eng.posterior(X, th, kernel, eng.pd_inverse(eng.gram(X, th, kernel)), f0, m)
    except Exception as e:      # a posterior precision that is not positive definite at this random theta: not the subject
        print(f"case {c}: skipped ({type(e).__name__})"); continue
    P0 = orc.posterior_covariance(Sinv0, f0, m, th[0])
    A0 = orc.variance_operator(Sinv0, P0, faithful=False, lam=orc.lambda_dense(f0, m, th[0]))
    Xc = rng.random((M, D))
    mu0, var0 = orc.predict_mean_var(Xc, X, th, Sinv0 @ f0, A0, kernel)
    o = eng.predict(post, Xc)
    e_mu = np.abs(host(o["mu"]) - mu0).max() / max(np.abs(mu0).max(), 1e-300)
    e_var = np.abs(host(o["var"]) - var0).max() / th[2] ** 2
    ok = e_mu < 1e-6 and e_var < 1e-6
    # a line batch big enough for the padded Y = G K* path and one below it
    msg = ""
    for B, G in ((40, 64), (3, 33)):
        al = np.sort(rng.random(G)); xis = np.zeros((B, D)); xis[np.arange(B), rng.integers(0, D, B)] = 1.0
        xs = rng.random((B, D)) * (xis == 0)
        grid = al[None, :, None] * xis[:, None, :] + xs[:, None, :]
        z = rng.standard_normal((64, G))
        ei, vm = eng.line_acq(post, grid, z, 0.0, jitter=1e-9 * th[2] ** 2)
        b = B // 2
        mu_b, cov_b = eng.predict_cov(post, grid[b])
        Ks = orc.cross_cov(X, grid[b], th, kernel)
        cov0 = orc.gram(grid[b], th, kernel) - Ks.T @ A0 @ Ks
        e_cov = np.abs(host(cov_b) - cov0).max() / th[2] ** 2
        e0 = orc.line_ei(host(mu_b), host(cov_b), z, 0.0, jitter=1e-9 * th[2] ** 2)
        e_ei = abs(host(ei)[b] - e0) / max(abs(e0), 1e-3 * th[2])
        ok = ok and e_cov < 1e-6 and e_ei < 1e-5
        msg += f" | B={B} G={G}: cov {e_cov:.1e} ei {e_ei:.1e}"
    worst = max(worst, e_mu, e_var)
    print(f"case {c}: m={m} n_q={n_q} N={N} D={D} M={M} {kernel} theta={np.round(th, 3)}: mu {e_mu:.1e} var {e_var:.1e}{msg} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        raise SystemExit(1)
print("fuzz ok, worst mean/variance error", worst)
