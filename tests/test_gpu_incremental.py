"""the incremental fit -- bordered Sigma^-1 when one query is appended
(feedback_processing.py:133-154) and the warm-started f_MAP -- against the full refit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine(0)


@pytest.mark.parametrize("name", ["smoke", "c2", "c3"])
def test_bordered_inverse_matches_full_inverse(eng, golden, name):
    """Append the fixture's queries one at a time from half the design: the bordered inverse must act like the
    full one (residual |Sigma Sigma^-1 - I| of the same size, Sigma^-1 f equal to the cond(Sigma) * eps floor)."""
    g = golden(name)
    X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
    mb = m + 1
    N = X.shape[0]
    n_q = N // mb
    q0 = max(1, n_q // 2)
    Xd = eng.dev(X)
    Sinv, Linv, L = eng.pd_inverse_factors3(eng.gram(Xd[:q0 * mb], th, kern))
    for q in range(q0, n_q):
        Sig = eng.gram(Xd[:(q + 1) * mb], th, kern)
        if q % 2:                                    # both entry points: with and without the bordered factor
            Sinv, Linv, L = eng.pd_inverse_append(Sig, Sinv, Linv, L)
        else:
            Sinv2, Linv2 = eng.pd_inverse_append(Sig, Sinv, Linv)
            Sinv, Linv, L = eng.pd_inverse_append(Sig, Sinv, Linv, L)
            assert np.array_equal(Sinv.cpu().numpy(), Sinv2.cpu().numpy())
    full, Lfull, Lfac = eng.pd_inverse_factors3(Sig)
    Lh, Lfh = np.tril(L.cpu().numpy()), np.tril(Lfac.cpu().numpy())
    assert np.abs(Lh - Lfh).max() <= 1e-8 * np.abs(Lfh).max()           # ... and is the factor of the new matrix
    assert np.abs(Lh @ Lh.T - Sig.cpu().numpy()).max() <= 1e-12 * np.abs(Lfh).max() ** 2
    Sg, A, B = Sig.cpu().numpy(), Sinv.cpu().numpy(), full.cpu().numpy()
    I = np.eye(N)
    res_app, res_full = np.abs(Sg @ A - I).max(), np.abs(Sg @ B - I).max()
    print(f"{name}: |Sigma Sinv - I| appended {res_app:.2e} / full {res_full:.2e} after {n_q - q0} appends")
    assert res_app <= 20 * res_full + 1e-9, (res_app, res_full)
    La, Lf = Linv.cpu().numpy(), Lfull.cpu().numpy()
    assert np.abs(np.triu(La, 1)).max() == 0.0
    assert np.abs(La - Lf).max() <= 1e-7 * np.abs(Lf).max()
    f = g["fMAP"]
    a_app, a_full = A @ f, B @ f
    assert np.abs(a_app - a_full).max() <= 1e-6 * np.abs(a_full).max()
    assert np.abs(A - A.T).max() <= 1e-9 * np.abs(A).max()
    # and the fit started from the same vector lands on the same f_MAP, on either path
    f1, _ = eng.fit_fmap(Sinv, g["f_init"], m, th[0], gtol=1e-6)
    f2, _ = eng.fit_fmap(full, g["f_init"], m, th[0], gtol=1e-6)
    assert np.abs(f1.cpu().numpy() - f2.cpu().numpy()).max() <= 1e-5 * np.abs(g["fMAP"]).max()
    f3, s3 = eng.fit_fmap(Sinv, g["f_init"], m, th[0], gtol=1e-6, L=L)
    assert s3["lbfgs_evals"] > 0 and np.abs(f3.cpu().numpy() - f2.cpu().numpy()).max() <= 1e-5 * np.abs(g["fMAP"]).max()


def test_append_rejects_bad_arguments(eng):
    A = eng.dev(np.eye(8))
    with pytest.raises(RuntimeError):
        eng.pd_inverse_append(A, eng.dev(np.eye(8)), eng.dev(np.eye(8)))          # nothing appended
    with pytest.raises(RuntimeError):
        eng.pd_inverse_append(eng.dev(np.eye(200)), eng.dev(np.eye(100)), eng.dev(np.eye(100)))   # > 64 rows at once


def test_append_reports_indefinite_border(eng):
    from ppbo_amd.engine import NotPositiveDefinite
    A = np.eye(6)
    A[5, 5] = -1.0
    with pytest.raises(NotPositiveDefinite):
        eng.pd_inverse_append(eng.dev(A), eng.dev(np.eye(4)), eng.dev(np.eye(4)))


def _replay(golden, incremental, method="whitened"):
    """Feed the reference's own C1 queries (fixture g7) to the drop-in, one at a time.  The global NumPy stream is
    re-seeded before every design update and every fit, so cold and incremental runs see identical designs."""
    from ppbo_amd.gp_model import GPModel
    from ppbo_amd.ppbo_settings import PPBO_settings
    g = golden("g7")
    st = PPBO_settings(D=2, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD", m=int(g["m"]),
                       theta_initial=list(map(float, g["theta"])), verbose=False)
    gp = GPModel(st, incremental=incremental)
    gp.fMAP_method = method
    n_init = int(g["n_init"])
    out = []
    for i in range(g["X_obs"].shape[0]):
        if i == n_init - 1:
            gp.turn_initialization_off()
        np.random.seed(1000 + i)
        gp.update_feedback_processing_object(g["X_obs"][:i + 1])
        gp.update_data()
        n_log = len(gp.fit_log)
        np.random.seed(2000 + i)
        gp.update_model()
        # Newton gap |P grad T| of this fit: how far SciPy's stopping rule (|grad| < 1e-4) leaves it from the optimum
        _, grad = gp.eng.T_and_grad(gp._dSigma_inv, gp.fMAP, gp.m, gp.theta[0])
        gap = float(np.abs(gp.posterior_covariance @ grad.cpu().numpy()).max()) if i >= n_init - 1 else np.inf
        out.append(dict(N=gp.N, fMAP=gp.fMAP.copy(), mustar=gp.mustar, xstar=gp.xstar.copy(), gap=gap,
                        chol=sum(t["n_cholesky"] for t in gp.fit_log[n_log:]),
                        evals=sum(t["lbfgs_evals"] for t in gp.fit_log[n_log:]),
                        iters=sum(t["iterations"] for t in gp.fit_log[n_log:]), X=gp.X.copy(),
                        n_appends=gp.n_appends, n_full=gp.n_full_inversions,
                        sinv_res=float(np.abs(gp.Sigma @ gp.Sigma_inv - np.eye(gp.N)).max())))
    return g, out


@pytest.mark.parametrize("method", ["whitened", "trust-region"])
def test_incremental_replay_matches_cold_refits(golden, method):
    """Same queries and designs, cold (a prior draw per update, the reference's default) vs incremental (bordered
    Sigma^-1 / L^-1 / L, warm start), both at the reference's stopping rule: the same f_MAP to 1e-5 max|f| plus the
    two fits' own Newton gaps, the same mu*, and markedly less work per query: >= 3x fewer factorizations on the
    trust-region path, >= 1.3x fewer O(N^2) evaluations (and still no factorization) on the whitened path (1.5x until the
    cold search learnt to open with the unit step: 23-33 evaluations per query cold against 16-21 warm now)."""
    g, cold = _replay(golden, False, method)
    g, inc = _replay(golden, True, method)
    n_init = int(g["n_init"])
    ratio, worst = [], 0.0
    # the model-level path really borders Sigma^-1 (one append per query after the first full inversion) ...
    assert cold[-1]["n_appends"] == 0 and cold[-1]["n_full"] == len(cold)
    assert inc[-1]["n_appends"] >= len(inc) - 2, (inc[-1]["n_appends"], inc[-1]["n_full"])
    assert inc[-1]["n_appends"] + inc[-1]["n_full"] == len(inc)
    # ... and the bordered inverse is as good an inverse as the full one
    for i in range(len(cold)):
        assert inc[i]["sinv_res"] <= 20 * cold[i]["sinv_res"] + 1e-9, (i, inc[i]["sinv_res"], cold[i]["sinv_res"])
    for i in range(n_init, len(cold)):
        assert np.array_equal(cold[i]["X"], inc[i]["X"])
        scale = np.abs(cold[i]["fMAP"]).max()
        df = np.abs(cold[i]["fMAP"] - inc[i]["fMAP"]).max()
        assert df <= 1e-5 * scale + 1.5 * (cold[i]["gap"] + inc[i]["gap"]), (i, cold[i]["N"], df, cold[i]["gap"], inc[i]["gap"])
        worst = max(worst, df / scale)
        assert abs(cold[i]["mustar"] - inc[i]["mustar"]) <= 1e-4 * max(abs(cold[i]["mustar"]), 1e-3)
        key = "chol" if method == "trust-region" else "evals"
        ratio.append(cold[i][key] / max(inc[i][key], 1))
    print(f"{method}: factorizations per query cold/incremental:", [(c["chol"], k["chol"]) for c, k in zip(cold[n_init:], inc[n_init:])])
    print(f"{method}: L-BFGS evaluations per query cold/incremental:", [(c["evals"], k["evals"]) for c, k in zip(cold[n_init:], inc[n_init:])])
    print(f"worst |f_cold - f_inc| / max|f| = {worst:.2e}")
    if method == "trust-region":
        assert np.median(ratio) >= 3.0, ratio
    else:
        assert np.median(ratio) >= 1.3, ratio
        assert max(k["chol"] for k in inc[n_init:]) <= 2


def test_incremental_switched_on_late_falls_back_to_full_inversion(golden):
    """incremental toggled on after a non-incremental update: no L^-1 exists yet, so the next update must do a
    full factor inversion (not call the append with a null factor), and the one after that borders."""
    from ppbo_amd.gp_model import GPModel
    from ppbo_amd.ppbo_settings import PPBO_settings
    g = golden("g7")
    st = PPBO_settings(D=2, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD", m=int(g["m"]),
                       theta_initial=list(map(float, g["theta"])), verbose=False)
    gp = GPModel(st, incremental=False)
    n_init = int(g["n_init"])
    for i in range(n_init + 3):
        if i == n_init - 1:
            gp.turn_initialization_off()
        if i == n_init:
            gp.incremental = True
        np.random.seed(1000 + i)
        gp.update_feedback_processing_object(g["X_obs"][:i + 1])
        gp.update_data()
        gp.update_model()
    assert gp.n_full_inversions == n_init + 1 and gp.n_appends == 2
    assert np.abs(gp.Sigma @ gp.Sigma_inv - np.eye(gp.N)).max() < 1e-6


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("seed", range(12))
def test_bordered_inverse_random_sizes(eng, seed):
    """Random leading size (1 .. 900, every residue of the 64-row panel) and border width (1 .. 64 rows) on random
    SPD matrices of random conditioning: inverse, L^-1 and L against LAPACK."""
    rng = np.random.default_rng(50 + seed)
    N1 = int(rng.integers(1, 901))
    k = int(rng.integers(1, 65))
    N = N1 + k
    Q = rng.standard_normal((N, N))
    A = Q @ Q.T / N + 10.0 ** rng.uniform(-3, 1) * np.eye(N)
    Ai1, Li1, L1 = eng.pd_inverse_factors3(A[:N1, :N1].copy())
    Ai, Li, L = eng.pd_inverse_append(A, Ai1, Li1, L1)
    ref = np.linalg.inv(A)
    Lref = np.linalg.cholesky(A)
    tag = f"N1={N1} k={k}"
    assert np.abs(host(Ai) - ref).max() <= 1e-9 * np.abs(ref).max(), tag
    assert np.abs(np.tril(host(L)) - Lref).max() <= 1e-10 * np.abs(Lref).max(), tag
    assert np.abs(np.tril(host(Li)) - np.linalg.inv(Lref)).max() <= 1e-9 * np.abs(np.linalg.inv(Lref)).max(), tag
    assert np.abs(host(Ai) @ A - np.eye(N)).max() <= 1e-8, tag
