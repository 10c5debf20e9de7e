"""The small operators of the path under the reference's own names -- GPModel.sum_Phi / sum_Phi_vec
(src/gp_model.py:176-218), misc.regularize_covariance / pd_inverse / is_positive_definite (src/misc.py:71-126), the
single-point acquisition objectives (src/acquisition.py:84-90, 109-113, 180-186) -- against outputs of the reference
itself (tests/golden/compat_<cfg>.npz, tools/make_golden_r3.py).  fp64 tolerance written at each assert."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine()


@pytest.mark.parametrize("name", ["smoke", "rq"])
def test_sum_phi_vs_reference(eng, golden, name):
    g, c = golden(name), load_golden(f"compat_{name}")
    m, sig = int(g["m"]), float(g["theta"][0])
    for tag, f in (("map", g["fMAP"]), ("init", g["f_init"])):
        for order in (0, 1, 2):
            ref = c[f"sum_phi_{tag}_{order}"]
            out = host(eng.sum_phi(f, m, sig, order))
            # order 0: the reference integrates with a 200-point Gauss-Hermite rule, the device uses the closed form
            assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (tag, order)
    with pytest.raises(RuntimeError, match="higher than 2"):
        eng.sum_phi(g["fMAP"], m, sig, 3)
    with pytest.raises(RuntimeError):
        eng.sum_phi(g["fMAP"][:-1], m, sig, 1)


@pytest.mark.parametrize("name", ["smoke", "rq"])
def test_gpmodel_sum_phi_surface(golden, name):
    from test_gpu_dropin import _model
    g, c = golden(name), load_golden(f"compat_{name}")
    gp, _ = _model(g)
    sig = float(g["theta"][0])
    for order in (0, 1, 2):
        v = gp.sum_Phi_vec(order, g["fMAP"], sig)
        assert v.shape == c[f"sum_phi_map_{order}"].shape
        assert np.abs(v - c[f"sum_phi_map_{order}"]).max() <= 1e-12 * max(1.0, np.abs(v).max())
    va = gp.sum_Phi_vec(1, g["f_init"], sig, over_all_indices=True)
    assert np.abs(va - c["sum_phi_all_init_1"]).max() <= 1e-12 * max(1.0, np.abs(va).max())
    for k, i in enumerate(c["sum_phi_rows"]):
        for order in (0, 1, 2):
            s = gp.sum_Phi(int(i), order, g["fMAP"], sig)
            assert abs(s - c["sum_phi_scalar"][k, order]) <= 1e-12 * max(1.0, abs(s))
    assert gp.sum_Phi_vec(3, g["fMAP"], sig) is None          # the reference prints and returns None (gp_model.py:203-204)


def test_regularize_covariance_vs_reference(eng):
    from ppbo_amd import misc
    c = load_golden("compat_smoke")
    for k in range(3):
        lev, pos = float(c[f"reg_arg_{k}"][0]), bool(c[f"reg_arg_{k}"][1])
        K = c["reg_in"].copy()
        out = misc.regularize_covariance(K, lev, pos)
        assert np.array_equal(K, c["reg_in"])                              # the argument of the wrapper is left alone
        # the reference's SVD round trip perturbs entries by a few ulp of |K|
        assert np.abs(out - c[f"reg_out_{k}"]).max() <= 1e-13 * np.abs(c["reg_in"]).max(), k
    # leading dimension > N and the error paths, through the engine
    import torch
    big = torch.zeros(40, 48, dtype=torch.float64, device=eng.device)
    big[:, :40] = torch.as_tensor(c["reg_in"], device=eng.device)
    view = big[:, :40]
    rc = eng.lib.ppbo_regularize_covariance(eng.ctx, view.data_ptr(), 40, 48, 1e-4, 1, 1e-7, eng._stream())
    assert rc == 0
    assert np.abs(host(view) - c["reg_out_0"]).max() <= 1e-13 * np.abs(c["reg_in"]).max()
    assert float(big[:, 40:].abs().max()) == 0.0                           # padding columns untouched
    with pytest.raises(RuntimeError, match="reg_level"):
        eng.regularize_covariance(c["reg_in"], 1.5)


def test_pd_inverse_and_definiteness_vs_reference(capsys):
    from ppbo_amd import misc
    from ppbo_amd.engine import NotPositiveDefinite
    c = load_golden("compat_smoke")
    reg = misc.regularize_covariance(c["pd_in"], 1e-4)
    assert np.abs(reg - c["pd_reg"]).max() <= 1e-13 * np.abs(c["pd_in"]).max()
    inv = misc.pd_inverse(c["pd_reg"])
    assert np.abs(inv - c["pd_inv"]).max() <= 1e-10 * np.abs(c["pd_inv"]).max()
    assert misc.is_positive_definite(c["pd_reg"]) is True and bool(c["is_pd"][0]) is True
    assert misc.is_positive_definite(c["reg_in"]) is False and bool(c["is_pd"][1]) is False
    assert "not positive definite" in capsys.readouterr().out               # the reference's message (misc.py:125)
    with pytest.raises(NotPositiveDefinite):
        misc.pd_inverse(c["reg_in"])


def test_single_point_objectives(golden):
    """EI_to_maximize / EI_fixed_x_to_maximize / varmax_to_maximize only re-arrange their argument and call EI / varmax
    (src/acquisition.py:84-90, 109-113, 180-186): same NumPy stream -> identical value, for the (1, D) array GPyOpt
    passes as well as for a flat vector."""
    from test_gpu_dropin import _model
    from ppbo_amd import acquisition as acq
    g = golden("smoke")
    gp, st = _model(g)
    gp.turn_initialization_off()
    np.random.seed(1)
    gp.update_model()
    D = gp.D
    xi_dims, x_dims = [0], [d for d in range(D) if d != 0]
    v = np.random.default_rng(0).random(D)
    xi, x = np.zeros(D), np.zeros(D)
    xi[xi_dims], x[x_dims] = v[xi_dims], v[x_dims]
    for wrapped, plain in ((acq.EI_to_maximize, acq.EI), (acq.varmax_to_maximize, acq.varmax)):
        np.random.seed(3); a = wrapped(v[None, :], xi_dims, x_dims, gp, 200)
        np.random.seed(3); b = wrapped(v, xi_dims, x_dims, gp, 200)
        np.random.seed(3); c0 = plain(xi, x, gp, 200)
        assert a == b == c0
    xs = np.asarray(gp.xstar, dtype=float)
    xi_ = xs.copy(); xi_[xi_dims] = 0.7
    np.random.seed(4); a = acq.EI_fixed_x_to_maximize(np.array([[0.7]]), xs, xi_dims, gp, 200)
    np.random.seed(4); b = acq.EI(xi_, xs, gp, 200)
    assert a == b and np.isfinite(a)


def test_device_normal_draws(eng):
    """ppbo_randn: a pure function of (seed, index) -- same seed same bits, a prefix of a longer request is the shorter
    one, different seeds differ -- with the moments and the distribution function of N(0, 1) (Kolmogorov-Smirnov at
    2^20 draws; the 1e-3 critical value is 1.95 / sqrt(n))."""
    from scipy.special import ndtr
    n = 1 << 20
    a = host(eng.randn(123, n))
    b = host(eng.randn(123, n))
    assert np.array_equal(a, b)
    assert np.array_equal(host(eng.randn(123, 1001)), a[:1001])           # odd length: the last pair is cut, not shifted
    c = host(eng.randn(124, n))
    assert not np.array_equal(a, c) and abs(np.corrcoef(a, c)[0, 1]) < 5e-3
    assert np.all(np.isfinite(a))
    assert abs(a.mean()) < 5.0 / np.sqrt(n) and abs(a.var() - 1.0) < 5.0 * np.sqrt(2.0 / n)
    assert abs((a ** 3).mean()) < 0.02 and abs((a ** 4).mean() - 3.0) < 0.05
    xs = np.sort(a)
    ks = np.abs(ndtr(xs) - (np.arange(1, n + 1) - 0.5) / n).max()
    assert ks < 1.95 / np.sqrt(n), ks
    assert abs(np.corrcoef(a[0::2], a[1::2])[0, 1]) < 5e-3                 # the two outputs of a Box-Muller pair
    assert abs(np.corrcoef(a[:-2], a[2:])[0, 1]) < 5e-3                    # neighbouring counters
    m = host(eng.randn(7, 300, 70))
    assert m.shape == (300, 70)
    with pytest.raises(RuntimeError):
        eng.lib.ppbo_randn.restype  # noqa: B018  (binding exists)
        eng._check(eng.lib.ppbo_randn(eng.ctx, 1, None, 10, eng._stream()), "ppbo_randn")


@pytest.mark.parametrize("F", [70, 1500, 3000, 4500])
def test_rff_omega_map_entry_point(eng, F):
    """ppbo_rff_omega_map on a small random basis: lands on a stationary point of S (|grad S| < gtol, every Hessian
    diagonal entry negative there: a maximum), a second call from the result is a no-op, bad sizes are refused.
    The feature counts cover every variant of the step kernel (vectors in 1, 2, 4 registers per thread; from memory)."""
    rng = np.random.default_rng(2)
    m, n_q = 5, 9
    N = n_q * (m + 1)
    Phi = rng.standard_normal((F, N)) * 0.2 * np.sqrt(70.0 / F)
    om, S, gn, it = eng.rff_omega_map(Phi, rng.standard_normal(F), m, 0.3, maxiter=500, gtol=1e-6)
    assert gn < 1e-6 and 0 < it < 500
    S1, g1, h1 = eng.rff_terms(Phi, om, m, 0.3)
    assert abs(S1 - S) <= 1e-12 * max(1.0, abs(S)) and np.linalg.norm(host(g1)) < 1e-6 and np.all(host(h1) < 0)
    om2, S2, gn2, it2 = eng.rff_omega_map(Phi, om, m, 0.3, maxiter=500, gtol=1e-6)
    assert it2 == 0 and np.array_equal(om2, om)
    # stopped by the iteration cap: the reported |grad S| (and S) belong to the point that is RETURNED, not to the one
    # the last step was computed from
    start = rng.standard_normal(F)
    for cap in (1, 2, 3, 5):
        omk, Sk, gnk, itk = eng.rff_omega_map(Phi, start, m, 0.3, maxiter=cap, gtol=1e-12)
        Sc, gc, _ = eng.rff_terms(Phi, omk, m, 0.3)
        assert itk == cap
        assert abs(Sc - Sk) <= 1e-12 * max(1.0, abs(Sk))
        assert abs(np.linalg.norm(host(gc)) - gnk) <= 1e-10 * max(1.0, gnk), (cap, np.linalg.norm(host(gc)), gnk)
    with pytest.raises(RuntimeError):
        eng.rff_omega_map(Phi[:, :-1], rng.standard_normal(F), m, 0.3)


def test_hsampler_sum_phi_vs_reference(golden):
    """Hsampler.sum_Phi / sum_Phi_vec (random_fourier_sampler.py:62-102) on the fixture's basis against the reference's
    own outputs: order 0 per query, orders 1 and 2 as nFeatures-vectors per query (<= 1e-12 of the largest entry)."""
    from test_gpu_dropin import _model
    from ppbo_amd.random_fourier_sampler import Hsampler
    g, c = golden("smoke"), load_golden("compat_smoke")
    gp, _ = _model(g)
    gp.set_theta()
    gp.xstar, gp.xstars_local = np.full(gp.D, 0.5), np.full((1, gp.D), 0.5)
    F = g["rff_W"].shape[0]
    hs = Hsampler(gp, F)
    hs.W, hs.b = g["rff_W"].copy(), g["rff_b"].reshape(F, 1).copy()
    hs.update_phi_X()
    sig = float(g["theta"][0])
    fw = hs.phi_X.T @ g["rff_omega"]
    assert np.abs(fw - c["hs_f"]).max() <= 1e-12 * np.abs(c["hs_f"]).max()
    for order in (0, 1, 2):
        out = hs.sum_Phi_vec(order, c["hs_f"], sig)
        ref = c[f"hs_sum_phi_{order}"]
        assert out.shape == ref.shape, (order, out.shape, ref.shape)
        assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), order
    one = hs.sum_Phi(int(hs.obs_indices[1]), 1, c["hs_f"], sig)
    assert np.abs(one - c["hs_sum_phi_1"][1]).max() <= 1e-12 * np.abs(c["hs_sum_phi_1"]).max()
    assert hs.sum_Phi(0, 3, c["hs_f"], sig) is None
