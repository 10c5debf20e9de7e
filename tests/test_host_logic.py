"""CPU-only checks of the host side of the drop-in (design shaping, settings, dispatcher bookkeeping)."""
import numpy as np
import pytest

from ppbo_amd.feedback_processing import FeedbackProcessing
from ppbo_amd.misc import alpha_bounds, hypercube_corners
from ppbo_amd.ppbo_settings import PPBO_settings


def _obs_rows(n_q, D, lo, hi, seed=0):
    rs = np.random.RandomState(seed)
    rows = []
    for q in range(n_q):
        d = q % D
        xi = np.zeros(D); xi[d] = 1.0
        x = lo + rs.rand(D) * (hi - lo); x[d] = 0.0
        a = lo[d] + rs.rand() * (hi[d] - lo[d])
        rows.append(np.concatenate([a * xi + x, xi, [a]]))
    return np.array(rows)


@pytest.mark.parametrize("dist", ["equispaced", "Cauchy", "TGN"])
def test_design_matrix_contract(dist):
    D, m = 3, 7
    bounds = ((-3, 3), (-2, 2), (0, 10))
    lo = np.array([b[0] for b in bounds], float); hi = np.array([b[1] for b in bounds], float)
    Xo = _obs_rows(5, D, lo, hi)
    np.random.seed(1)
    fp = FeedbackProcessing(D, m, bounds, dist, 0.4)
    fp.initialize_data(Xo[:3])
    fp.update_data(Xo[:4])
    fp.update_data(Xo[:5])
    assert fp.N == 5 * (m + 1) and fp.X.shape == (fp.N, D)
    assert fp.obs_indices == [q * (m + 1) for q in range(5)]
    assert fp.latest_obs_indices == [(i // (m + 1)) * (m + 1) for i in range(fp.N)]
    assert len(fp.pseudobs_indices) == 5 * m
    assert fp.X.min() >= -1e-12 and fp.X.max() <= 1 + 1e-12
    for q in range(5):
        blk = fp.X[q * (m + 1):(q + 1) * (m + 1)]
        d = q % D
        assert np.allclose(blk[0], fp.scale(Xo[q, :D]))
        others = [k for k in range(D) if k != d]
        assert np.allclose(blk[1:, others], blk[0, others])          # pseudo-observations stay on the line
        assert len(np.unique(blk[1:, d])) == m                         # m distinct grid points
    assert np.allclose(fp.unscale(fp.scale(Xo[:, :D])), Xo[:, :D])
    z = np.array([[0.0, 1.0, 0.0]])
    assert np.array_equal(fp.unscale(z, retain_0_values=True) == 0, z == 0)


def test_xi_grid_scaled_line():
    fp = FeedbackProcessing(4, 5, ((0, 1),) * 4, "equispaced", 0.4)
    np.random.seed(0)
    g = fp.xi_grid(xi=np.array([0, 1.0, 0, 0]), x=np.array([0.2, 0.0, 0.4, 0.6]), m=70, is_scaled=True)
    assert g.shape == (70, 4) and np.all(np.diff(g[:, 1]) > 0) and g[:, 1].min() >= 0 and g[:, 1].max() <= 1
    assert np.allclose(g[:, [0, 2, 3]], [0.2, 0.4, 0.6])


def test_alpha_bounds_and_corners():
    lo, hi = alpha_bounds([1.0, 0, -0.5], [-3, -2, -1], [3, 2, 1])
    assert np.isclose(lo, -2.0) and np.isclose(hi, 2.0)
    c = hypercube_corners(((-3, 3), (-2, 2)))
    assert c.shape == (4, 2) and {tuple(r) for r in c} == {(-3, -2), (-3, 2), (3, -2), (3, 2)}


def test_settings_derivations():
    s = PPBO_settings(D=6, bounds=((0, 1),) * 6, xi_acquisition_function="PCD", m=31, verbose=False)
    assert s.dim_query_prev_iter == 6 and s.x_acquisition_function == "exploit" and s.n_pseudoobservations == 31
    s = PPBO_settings(D=6, bounds=((0, 1),) * 6, xi_acquisition_function="EI", verbose=False)
    assert s.xi_dims_prev_iter == [0, 1] and s.x_acquisition_function == "none"
    s = PPBO_settings(D=2, bounds=((0, 1),) * 2, xi_acquisition_function="EXR", verbose=False)
    assert s.xi_dims_prev_iter == [1]
    s = PPBO_settings(D=3, bounds=((0, 1),) * 3, xi_acquisition_function="COORDINATE-VARMAX", verbose=False)
    assert s.x_acquisition_function == "varmax" and s.dim_query_prev_iter == 3
    assert s.fMAP_optimizer == "trust-exact" and s.mc_samples == 150 and s.n_gausshermite_sample_points == 200


def test_polish_host_logic():
    """GPModel._polish (the host half of mu_star's refinement: bounded L-BFGS-B on the device's mean + gradient)
    driven by a stand-in engine whose mean_grad is the oracle's: it must not lose value and must end on a
    stationary point of mu inside the box.  (The device-resident ascent and search are checked in
    tests/test_gpu_mean_search.py.)"""
    import torch
    from oracle import ppbo_oracle as orc
    from ppbo_amd.gp_model import GPModel

    rng = np.random.default_rng(3)
    D, N = 3, 40
    X = rng.random((N, D))
    theta = np.array([0.1, 0.35, 1.0])
    alpha = rng.standard_normal(N)

    class Eng:
        def mean_grad(self, post, Xc):
            mu, g = orc.mean_grad(np.asarray(Xc, dtype=float), X, theta, alpha, "SE_kernel")
            return torch.from_numpy(mu), torch.from_numpy(g)

    gp = object.__new__(GPModel)
    gp.eng, gp.D, gp.bounds, gp._post_mean = Eng(), D, ((0, 1),) * D, object()
    starts = rng.random((12, D))
    for _ in range(200):            # the polish is applied to the ascent's output: get near a maximum first
        _, g0 = orc.mean_grad(starts, X, theta, alpha)
        starts = np.clip(starts + 0.002 * g0 / max(np.abs(g0).max(), 1e-300) * 5.0, 0.0, 1.0)
    mu0, _ = orc.mean_grad(starts, X, theta, alpha)
    for k in range(len(starts)):
        xp, vp = gp._polish(starts[k])
        assert vp >= mu0[k] - 1e-12 and np.all((xp >= 0) & (xp <= 1))
        _, gp_ = orc.mean_grad(xp[None, :], X, theta, alpha)
        pgp = np.where(((xp <= 0) & (gp_[0] < 0)) | ((xp >= 1) & (gp_[0] > 0)), 0.0, gp_[0])
        assert np.abs(pgp).max() < 1e-5 * max(1.0, np.abs(alpha).max())


# ---- the next_query dispatcher against the reference's own outputs (tools/make_golden_r2.py) -----------------
@pytest.mark.parametrize("name", ["smoke", "rq", "c2", "c4", "c3"])
@pytest.mark.parametrize("acq,xacq", [("PCD", "exploit"), ("EXT", "exploit"), ("RAND", "exploit"), ("RAND", "random"),
                                      ("PCD", "random")])
def test_next_query_matches_reference_outputs(name, acq, xacq):
    """acquisition.py:9-65 with the reference's x* and a seeded global stream: identical xi, x (unscaled, zeros
    retained, xi normalised by its max) and identical dim_query_prev_iter bookkeeping, call after call (the PCD /
    EXT cycle wraps after D calls).  No GPU: these strategies only read GP_model.xstar and FP.unscale."""
    import os
    from types import SimpleNamespace
    from conftest import GOLDEN, load_golden
    if not os.path.exists(os.path.join(GOLDEN, f"{name}_x.npz")):
        pytest.skip(f"{name}_x.npz not generated")
    from ppbo_amd.acquisition import next_query
    x, g = load_golden(name + "_x"), load_golden(name)
    D = int(g["D"])
    bounds = tuple(map(tuple, g["bounds"]))
    st = PPBO_settings(D=D, bounds=bounds, xi_acquisition_function=acq, theta_initial=list(g["theta"]), m=int(g["m"]),
                       verbose=False, kernel=str(g["kernel"]))
    st.x_acquisition_function = xacq
    fp = FeedbackProcessing(D, int(g["m"]), bounds, "equispaced", 0.4)
    gp = SimpleNamespace(xstar=x["xstar"].copy(), FP=fp, verbose=False, D=D)
    key = f"nq_{acq}_{xacq}"
    for k in range(x[key + "_xi"].shape[0]):
        np.random.seed(500 + k)
        xi, xx = next_query(st, gp, unscale=True)
        assert np.abs(xi - x[key + "_xi"][k]).max() <= 1e-12 * max(1.0, np.abs(x[key + "_xi"][k]).max()), (k, xi)
        assert np.abs(xx - x[key + "_x"][k]).max() <= 1e-12 * max(1.0, np.abs(x[key + "_x"][k]).max()), (k, xx)
        assert np.array_equal(xi == 0, x[key + "_xi"][k] == 0) and np.array_equal(xx == 0, x[key + "_x"][k] == 0)
        assert int(getattr(st, "dim_query_prev_iter", -1)) == int(x[key + "_dim"][k])
    assert np.array_equal(gp.xstar, x["xstar"])                 # the dispatcher must not mutate the model's x*


def test_tgn_sampler_follows_the_reference_density():
    """TGN grids (feedback_processing.py:87-95): the reference draws them by adaptive rejection sampling (arspy,
    absent) from log_TGN_pdf (TGN_distribution.py:21-25).  tests/golden/tgn.npz holds that log-density evaluated by the
    reference itself; our inverse-CDF sampler must (i) use exactly that density and (ii) produce samples whose
    empirical distribution matches it (Kolmogorov-Smirnov)."""
    import scipy.stats
    from scipy.special import gamma as Gamma
    from conftest import load_golden
    from ppbo_amd.feedback_processing import _tgn_sample
    g = load_golden("tgn")
    rng = np.random.RandomState(0)
    for (gam, al, a, b), x, lp in zip(g["cases"], g["x"], g["logpdf"]):
        scale = Gamma(gam) * abs(b - a) / 10.0
        dist = scipy.stats.gennorm(gam, loc=al, scale=scale)
        mass = dist.cdf(b) - dist.cdf(a)
        ours = dist.logpdf(x) - np.log(mass)                       # the density _tgn_sample inverts
        assert np.abs(ours - lp).max() <= 1e-9 * max(1.0, np.abs(lp).max())
        s = _tgn_sample(20000, gam, al, a, b, rng=rng)
        assert s.min() >= a and s.max() <= b
        cdf = lambda v: (dist.cdf(v) - dist.cdf(a)) / mass         # noqa: E731
        assert scipy.stats.kstest(s, cdf).pvalue > 1e-3


def test_run_ppbo_loop_call_order_matches_the_reference(monkeypatch):
    """ppbo_numerical_main.py:57-127 restated in ppbo_amd/numerical_main.py: the sequence of model calls (creation at the
    first query, turn_initialization_off before the LAST initial query's update and again after the block,
    mustar_previous_iteration, optimize_theta at the configured query, x zeroed on xi's support) -- checked on CPU
    with a recording stand-in for GPModel."""
    from ppbo_amd import numerical_main as nm
    log = []

    class FakeFP:
        def unscale(self, v, retain_0_values=False):
            return np.asarray(v, dtype=float)

    class FakeModel:
        def __init__(self, settings, engine=None, incremental=False):
            log.append(("create", incremental))
            self.D = settings.D
            self.FP = FakeFP()
            self.xstar = np.full(settings.D, 0.25)
            self.mustar = 0.0
            self.verbose = False
            self.mustar_previous_iteration = None
            self.n = 0
        def turn_initialization_off(self): log.append(("init_off", self.n))
        def set_last_iteration(self): log.append(("last", self.n))
        def update_feedback_processing_object(self, X_obs): self.n = len(X_obs); log.append(("fp", self.n, X_obs[-1].copy()))
        def update_data(self): log.append(("data", self.n))
        def update_model(self, optimize_theta=False):
            log.append(("model", self.n, bool(optimize_theta)))
            self.mustar = float(self.n)

    monkeypatch.setattr(nm, "GPModel", FakeModel)
    D = 3
    st = PPBO_settings(D=D, bounds=((0, 1),) * D, xi_acquisition_function="PCD", m=5, verbose=False)
    xi0 = np.eye(D)
    x0 = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]])
    res, xs, ms, gp = nm.run_ppbo_loop(lambda xi, x: 0.5, xi0, x0, 4, st, optimize_hyperparameters_after_actual_query_number=2)
    kinds = [e[0] for e in log]
    assert kinds[0] == "create" and kinds.count("create") == 1
    # initial block: fp/data/model per query; init_off right before the third (last) initial update and after the block
    assert [e for e in log if e[0] == "init_off"] == [("init_off", 2), ("init_off", 3)]
    assert [e[1:] for e in log if e[0] == "model"] == [(1, False), (2, False), (3, False), (4, False), (5, True), (6, False), (7, False)]
    assert not any(k == "last" for k in kinds)                       # :104 never fires with initial queries
    # x is zeroed on xi's support and the stored row is [alpha*xi + x ; xi ; alpha]
    assert np.allclose(res[0], [0.5, 0.2, 0.3, 1, 0, 0, 0.5]) and np.allclose(res[1], [0.4, 0.5, 0.6, 0, 1, 0, 0.5])
    # actual queries: PCD directions cycle 1, 2, 3, 1 and x exploits the model's x* off the direction
    assert [int(np.argmax(r[D:2 * D])) for r in res[3:]] == [0, 1, 2, 0]
    assert np.allclose(res[3], [0.5, 0.25, 0.25, 1, 0, 0, 0.5])
    assert gp.mustar_previous_iteration == 6.0                       # the model's mu* BEFORE the last update (:113)
    assert res.shape == (7, 2 * D + 1) and xs.shape == (7, D) and ms == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    # adaptive initialisation (:74-75) copies the previous answer into the remaining initial x's
    log.clear()
    res2, *_ = nm.run_ppbo_loop(lambda xi, x: 0.5, xi0, x0.copy(), 0, st, adaptive_initialization=True)
    assert np.allclose(res2[1], [0.5, 0.5, 0.3, 0, 1, 0, 0.5])       # x := previous chosen point (0.5, 0.2, 0.3), then its 2nd coord zeroed


def test_create_X_and_update_X_agree():
    """src/feedback_processing.py:110-154: update_X appends the block of the newest row of X_obs to what create_X built
    from the earlier rows; both routes draw one jittered grid per row in row order, so from the same NumPy seed they
    give the same design."""
    from ppbo_amd.feedback_processing import FeedbackProcessing
    D, m = 3, 6
    bounds = ((-1.0, 2.0), (0.0, 1.0), (-3.0, 3.0))
    rng = np.random.default_rng(5)
    rows = []
    for q in range(5):
        xi = np.zeros(D); xi[q % D] = 1.0
        x = np.array([rng.uniform(lo, hi) for lo, hi in bounds]); x[q % D] = 0.0
        a = rng.uniform(*bounds[q % D])
        rows.append(list(a * xi + x) + list(xi) + [a])
    X_obs = np.array(rows)
    one = FeedbackProcessing(D, m, bounds, "equispaced", None)
    np.random.seed(8)
    one.initialize_data(X_obs)
    two = FeedbackProcessing(D, m, bounds, "equispaced", None)
    np.random.seed(8)
    two.initialize_data(X_obs[:3])
    for k in (4, 5):
        two.update_data(X_obs[:k])
    assert two.iter_number == one.iter_number + 2
    assert np.array_equal(one.X_full, two.X_full) and np.array_equal(one.X, two.X) and one.N == two.N == 5 * (m + 1)
    assert list(one.obs_indices) == list(two.obs_indices) == [q * (m + 1) for q in range(5)]
    three = FeedbackProcessing(D, m, bounds, "equispaced", None)
    three.X_obs = X_obs
    np.random.seed(8)
    three.create_X()                                    # the reference's two-step form (:37-38)
    three.create_indices_bookkeeping()
    assert np.array_equal(three.X, one.X)


# ---- a-15: the objectives the outer searches score are the reference's own (no GPU: the line scorer is a stub) --------
class _StubEngine:
    def randn(self, seed, *shape):
        return np.zeros(shape)


class _StubModel:
    def __init__(self, D):
        self.D, self.eng = D, _StubEngine()
        self.xstar = np.linspace(0.15, 0.85, D)
        self.mustar, self.theta = 0.0, [0.1, 0.3, 0.5]


def _recording_scorer(monkeypatch):
    from ppbo_amd import acquisition as acq
    seen = []

    def fake(xis, xs, GP_model, mc_samples, z=None, alphas=None):
        xis, xs = np.atleast_2d(xis), np.atleast_2d(xs)
        seen.append((xis.copy(), xs.copy()))
        v = -((xis - 0.3) ** 2).sum(axis=1) - ((xs - 0.6) ** 2).sum(axis=1)        # any smooth landscape
        return v, v

    monkeypatch.setattr(acq, "_line_scores", fake)
    return seen


def test_maximize_EI_fixed_x_scores_the_references_objective(monkeypatch):
    """src/acquisition.py:109-131: BO varies xi on xi_dims only; the objective is EI(xi_, xstar) with xi_ = xstar.copy(),
    xi_[xi_dims] = candidate, and the FULL xstar as x.  The pair that is RETURNED is a different line: xi zero off
    xi_dims, x = xstar off xi_dims and zero on them."""
    from ppbo_amd import acquisition as acq
    from ppbo_amd.ppbo_settings import PPBO_settings
    D, xi_dims = 5, [1, 3]
    gp = _StubModel(D)
    st = PPBO_settings(D=D, bounds=((0, 1),) * D, xi_acquisition_function="EI-FIXEDX", verbose=False)
    seen = _recording_scorer(monkeypatch)
    np.random.seed(0)
    xi, x = acq.maximize_EI_fixed_x(xi_dims, gp, st)
    x_dims = [0, 2, 4]
    assert seen
    for xis, xs in seen:
        assert np.array_equal(xs, np.tile(gp.xstar, (len(xs), 1)))                  # x = xstar, every coordinate
        assert np.array_equal(xis[:, x_dims], np.tile(gp.xstar[x_dims], (len(xis), 1)))   # xi keeps xstar off xi_dims
        assert np.all((xis[:, xi_dims] >= 0) & (xis[:, xi_dims] <= 1))
    assert np.all(xi[x_dims] == 0) and np.all(x[xi_dims] == 0) and np.allclose(x[x_dims], gp.xstar[x_dims])
    assert np.allclose(gp.acq_search_scored[0][xi_dims], xi[xi_dims], atol=1e-6)


def test_maximize_varmax_given_xi_searches_the_whole_box(monkeypatch):
    """src/acquisition.py:208-218: the objective is varmax(xi, x) over ALL D coordinates of x; the coordinates on xi's
    support are zeroed in the result, not in the search."""
    from ppbo_amd import acquisition as acq
    from ppbo_amd.ppbo_settings import PPBO_settings
    D = 4
    gp = _StubModel(D)
    st = PPBO_settings(D=D, bounds=((0, 1),) * D, xi_acquisition_function="COORDINATE-VARMAX", verbose=False)
    xi = np.array([0.0, 1.0, 0.0, 0.0])
    seen = _recording_scorer(monkeypatch)
    np.random.seed(1)
    x = acq.maximize_varmax_given_xi(xi, gp, st)
    assert any(np.any(xs[:, 1] != 0.0) for _, xs in seen)                          # x varies on xi's support too
    assert all(np.array_equal(xis, np.tile(xi, (len(xis), 1))) for xis, _ in seen)
    assert x[1] == 0.0 and np.all(x[[0, 2, 3]] > 0)
    assert gp.acq_search_scored[1][1] != 0.0 and np.array_equal(gp.acq_search_scored[1][[0, 2, 3]], x[[0, 2, 3]])


def test_noisy_alpha_rows_follow_the_reference_recipe():
    """70 abscissae per line: linspace(0.005, 0.995) + N(0, 0.01), clipped to [0, 1], sorted, all distinct
    (src/feedback_processing.py:57-74 with is_scaled); B lines in one call consume the stream like B calls of one."""
    from ppbo_amd import acquisition as acq
    np.random.seed(4)
    a = acq._noisy_alpha_rows(300)
    assert a.shape == (300, 70) and np.all(np.diff(a, axis=1) > 0) and a.min() >= 0.0 and a.max() <= 1.0
    assert np.abs(a - np.linspace(0.005, 0.995, 70)).max() < 0.06
    # the first pass is ONE block of B x 70 normals, row b = what the b-th sequential xi_grid call would have drawn;
    # only rows in which two values were clipped onto the same boundary (a redraw: ~2 % of the rows) differ
    np.random.seed(4)
    first = np.sort(np.clip(np.linspace(0.005, 0.995, 70) + np.random.normal(0.0, 0.01, (300, 70)), 0.0, 1.0), axis=1)
    clean = ~(np.diff(first, axis=1) == 0.0).any(axis=1)
    assert clean.sum() >= 280 and np.array_equal(a[clean], first[clean])
    np.random.seed(4)
    k = int(np.argmin(clean)) if not clean.all() else 300          # sequential calls agree up to the first redraw
    b = np.stack([acq._noisy_alpha_rows(1)[0] for _ in range(k)])
    assert np.array_equal(a[:k], b)
