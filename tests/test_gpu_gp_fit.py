"""ppbo_gp_fit: one whole GP fit in one library call (Sigma, Cholesky factor, Sigma^-1 on a side stream, the whitened
f_MAP search steered through a host-mapped progress word, the posterior) against the same work done call by call and
against the reference's fixtures (src/gp_model.py:91-117)."""
import numpy as np
import pytest

from conftest import golden_names

pytestmark = pytest.mark.gpu
FITTED = golden_names(lambda n: n != "c5")


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine(0)


@pytest.mark.parametrize("name", FITTED)
def test_gp_fit_equals_the_separate_calls_and_the_reference(eng, golden, name):
    g = golden(name)
    X, th, kern, m, sig = g["X"], g["theta"], str(g["kernel"]), int(g["m"]), float(g["theta"][0])
    r = eng.gp_fit(X, th, kern, m, g["f_init"], gtol=1e-4, want_Linv=True)
    st = r["stats"]
    print(name, st)
    # the matrices come from the same kernels as the separate calls: the same bits
    S = eng.gram(X, th, kern)
    Sinv, L = eng.pd_inverse_chol(S)
    assert np.array_equal(host(r["Sigma"]), host(S))
    assert np.array_equal(np.tril(host(r["L"])), np.tril(host(L)))
    assert np.array_equal(host(r["Sigma_inv"]), host(Sinv))
    Li = np.tril(host(r["Linv"]))
    assert np.abs(Li @ np.tril(host(L)) - np.eye(X.shape[0])).max() <= 1e-7
    # f_MAP: the reference's optimum, by the reference's rule
    f = host(r["fMAP"])
    assert st["converged"] and st["gradnorm"] < 1e-4 and st["lbfgs_evals"] > 0
    T, grad = eng.T_and_grad(Sinv, f, m, sig)
    assert abs(T - st["T"]) <= 1e-8 * max(1.0, abs(T)) and np.linalg.norm(host(grad)) < 1e-4   # phi(z) vs T(f): cond(Sigma) ~ 1e7 apart in rounding
    assert abs(np.linalg.norm(host(grad)) - st["gradnorm"]) <= 1e-6 * max(1.0, st["gradnorm"]) + 1e-9
    post_ref = eng.posterior(X, th, kern, Sinv, g["fMAP"], m, want_P=True)
    P = host(post_ref.P)
    gaps = 0.0
    for fv in (f, g["fMAP"]):
        _, gr = eng.T_and_grad(Sinv, fv, m, sig)
        gaps += np.abs(P @ host(gr)).max()
    assert np.abs(f - g["fMAP"]).max() <= 1e-5 * np.abs(g["fMAP"]).max() + 1.5 * gaps
    assert st["T"] >= float(g["T_fMAP"]) - 1e-7 * max(1.0, abs(float(g["T_fMAP"])))
    # the posterior state is ppbo_posterior's at that f_MAP, bit for bit
    post2 = eng.posterior(X, th, kern, Sinv, r["fMAP"], m)
    for a, b in ((r["post"].alpha, post2.alpha), (r["post"].lam_diag, post2.lam_diag), (r["post"].lam_off, post2.lam_off),
                 (r["post"].G, post2.G)):
        assert np.array_equal(host(a), host(b))
    # and it predicts like the reference on the fixture's candidates, up to what two f_MAPs that both stop at
    # |grad| < 1e-4 may differ by (the Newton gaps above; the 1e-6 parity of mean / variance at a FIXED f_MAP is
    # test_gpu_parity.py's business)
    o = eng.predict(r["post"], g["Xc"], want_best=False)
    sf2 = float(th[2]) ** 2
    assert np.abs(host(o["mu"]) - g["mu"]).max() <= 2e-4 * np.abs(g["mu"]).max()
    assert np.abs(host(o["var"]) - g["var"]).max() <= 2e-4 * sf2


@pytest.mark.parametrize("name", ["smoke", "c2"])
def test_gp_fit_from_a_whitened_start(eng, golden, name):
    """start_is_whitened: the start is the prior draw L z0 (src/gp_model.py:374,381) given by its z0."""
    g = golden(name)
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    z0 = np.random.default_rng(3).standard_normal(X.shape[0])
    a = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
    f0 = eng.dgemv(a["L"], z0, lower=True)
    b = eng.gp_fit(X, th, kern, m, f0)
    assert a["stats"]["converged"] and b["stats"]["converged"]
    assert abs(a["stats"]["T"] - b["stats"]["T"]) <= 1e-6 * max(1.0, abs(b["stats"]["T"]))
    assert np.abs(host(a["fMAP"]) - host(b["fMAP"])).max() <= 1e-4 * np.abs(host(b["fMAP"])).max()


def test_gp_fit_second_stream_changes_nothing(eng, golden, monkeypatch):
    """From N = 1024 on a whitened start lets ppbo_gp_fit form L^-1 and Sigma^-1 on the ctx's second stream beside the
    first evaluations of the search (PPBO_FIT_OVERLAP, default on).  The search's stream joins them at a FIXED slot, so
    the result is bitwise the one of a ctx that does everything on one stream, call after call; a budget that ends the
    search before the join slot still hands the finisher a finished Sigma^-1."""
    from ppbo_amd.engine import Engine
    g = golden("c3")
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    z0 = np.random.default_rng(3).standard_normal(X.shape[0])
    a = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
    a2 = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
    monkeypatch.setenv("PPBO_FIT_OVERLAP", "0")
    plain = Engine(0)
    try:
        b = plain.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
        assert a["stats"] == b["stats"] == a2["stats"] and a["stats"]["lbfgs_evals"] > 8
        for key in ("fMAP", "Sigma_inv", "L"):
            assert np.array_equal(host(a[key]), host(b[key])) and np.array_equal(host(a[key]), host(a2[key])), key
        assert np.array_equal(host(a["post"].G), host(b["post"].G)) and np.array_equal(host(a["post"].alpha), host(b["post"].alpha))
        c = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True, lbfgs_max_evals=5)
        d = plain.gp_fit(X, th, kern, m, z0, start_is_whitened=True, lbfgs_max_evals=5)
        assert c["stats"]["lbfgs_status"] == 5 and c["stats"]["converged"] and c["stats"] == d["stats"]
        assert np.array_equal(host(c["fMAP"]), host(d["fMAP"]))
        assert np.abs(host(c["fMAP"]) - host(a["fMAP"])).max() <= 1e-4 * np.abs(host(a["fMAP"])).max()
    finally:
        plain.close()


def test_gp_fit_is_deterministic_and_reusable(eng, golden):
    g = golden("c2")
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    a = eng.gp_fit(X, th, kern, m, g["f_init"])
    b = eng.gp_fit(X, th, kern, m, g["f_init"])
    assert a["stats"] == b["stats"]
    assert np.array_equal(host(a["fMAP"]), host(b["fMAP"])) and np.array_equal(host(a["post"].G), host(b["post"].G))
    c = eng.gp_fit(X, th, kern, m, g["f_init"], want_posterior=False, want_Sigma=False)
    assert c["post"] is None and c["Sigma"] is None and np.array_equal(host(c["fMAP"]), host(a["fMAP"]))
    # a tiny evaluation budget: the trust-region finisher takes over inside the same call
    d = eng.gp_fit(X, th, kern, m, g["f_init"], lbfgs_max_evals=6)
    assert d["stats"]["lbfgs_status"] == 5 and d["stats"]["converged"] and d["stats"]["n_cholesky"] > 0
    assert np.abs(host(d["fMAP"]) - host(a["fMAP"])).max() <= 1e-4 * np.abs(host(a["fMAP"])).max()


def test_gp_fit_reports_a_posterior_that_is_not_positive_definite(eng, golden):
    """A start far from any maximum and a budget of zero useful work cannot produce this; what does is handing the
    call a sigma so small that Lambda dominates -- the call must say info = 2, keep f_MAP, and not raise."""
    g = golden("smoke")
    X, kern, m = g["X"], str(g["kernel"]), int(g["m"])
    with pytest.raises(ValueError):
        eng.gp_fit(X, g["theta"], kern, m, np.zeros(X.shape[0] + 1))
    with pytest.raises(RuntimeError):
        eng.gp_fit(X, [0.0, 0.3, 0.5], kern, m, g["f_init"])          # sigma must be positive


@pytest.mark.parametrize("name,whitened", [("smoke", False), ("smoke", True), ("c3", False), ("c3", True)])
def test_gp_fit_says_so_at_once_when_sigma_is_not_positive_definite(eng, golden, name, whitened):
    """A negative shrinkage pushes Sigma's small eigenvalues below zero: potrf stops early and leaves a half-factored
    matrix.  The search's first launch reads the info word on the device and ends the search before its first
    evaluation, so the call returns PPBO_ERR_NOT_PD with info = 1 -- not whatever the pipeline would have made of the
    garbage -- and the ctx is as good as new afterwards."""
    from ppbo_amd.engine import NotPositiveDefinite
    g = golden(name)
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    z0 = np.random.default_rng(3).standard_normal(X.shape[0])
    start = z0 if whitened else g["f_init"]
    good = eng.gp_fit(X, th, kern, m, start, start_is_whitened=whitened)
    with pytest.raises(NotPositiveDefinite) as e:
        eng.gp_fit(X, th, kern, m, start, shrink=-0.5, start_is_whitened=whitened)
    assert e.value.info == 1 and "Sigma is not positive definite" in str(e.value)
    again = eng.gp_fit(X, th, kern, m, start, start_is_whitened=whitened)
    assert again["stats"] == good["stats"] and np.array_equal(host(again["fMAP"]), host(good["fMAP"]))


def test_whitened_search_stops_on_a_missing_factor(eng, golden):
    """The standalone entry cannot see a factorization's info word (it is handed L); ppbo_gp_fit hands it over, and
    status 6 is what the search reports then: zero evaluations."""
    g = golden("smoke")
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    from ppbo_amd.engine import NotPositiveDefinite
    with pytest.raises(NotPositiveDefinite):
        eng.gp_fit(X, th, kern, m, g["f_init"], shrink=-0.5, want_posterior=False, want_Sigma=False)


def test_gp_model_raises_not_positive_definite_through_the_fused_fit(eng, golden):
    from ppbo_amd.engine import NotPositiveDefinite
    from test_gpu_dropin import _model
    g = golden("smoke")
    model, _ = _model(g)
    model.COVARIANCE_SHRINKAGE = -0.5
    np.random.seed(0)
    with pytest.raises(NotPositiveDefinite):
        model.update_model()


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_a_stalled_progress_word_is_not_an_error(golden, monkeypatch, name):
    """PPBO_POLL_LIMIT_MS = 0: every time the host has nothing to enqueue and the progress word has been still for
    ~2000 polls it falls back to hipStreamSynchronize -- the path a slow or shared device takes after five seconds.
    The search must carry on from there with more slots (it used to report 'made no progress') and end
    exactly where the undisturbed one does."""
    from ppbo_amd.engine import Engine, get_engine
    g = golden(name)
    X, th, kern, m = g["X"], g["theta"], str(g["kernel"]), int(g["m"])
    ref = get_engine(0).gp_fit(X, th, kern, m, g["f_init"])
    monkeypatch.setenv("PPBO_POLL_LIMIT_MS", "0")
    slow = Engine(0)
    try:
        r = slow.gp_fit(X, th, kern, m, g["f_init"])
        assert r["stats"] == ref["stats"] and np.array_equal(host(r["fMAP"]), host(ref["fMAP"]))
        f, st = slow.fit_fmap(r["Sigma_inv"], g["f_init"], m, float(th[0]), L=r["L"])
        f2, st2 = get_engine(0).fit_fmap(ref["Sigma_inv"], g["f_init"], m, float(th[0]), L=ref["L"])
        assert st == st2 and np.array_equal(host(f), host(f2))
    finally:
        slow.close()


def test_a_stalled_progress_word_is_not_an_error_for_omega_map(monkeypatch):
    """The same for ppbo_rff_omega_map (several hundred slots): same iterate count, same point, bit for bit."""
    from ppbo_amd.engine import Engine, get_engine
    rng = np.random.default_rng(2)
    m, n_q, F = 5, 9, 1500
    N = n_q * (m + 1)
    Phi = rng.standard_normal((F, N)) * 0.2 * np.sqrt(70.0 / F)
    w0 = rng.standard_normal(F)
    om, S, gn, it = get_engine(0).rff_omega_map(Phi, w0, m, 0.3, maxiter=500, gtol=1e-6)
    monkeypatch.setenv("PPBO_POLL_LIMIT_MS", "0")
    slow = Engine(0)
    try:
        om2, S2, gn2, it2 = slow.rff_omega_map(Phi, w0, m, 0.3, maxiter=500, gtol=1e-6)
        assert it2 == it and S2 == S and gn2 == gn and np.array_equal(om2, om)
    finally:
        slow.close()
