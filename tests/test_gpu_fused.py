"""GPU: the one-launch scoring kernel (csrc/fused.hip: K* tile in LDS, contraction, score, block argmax) against the
three-launch form (kstar -> quadform -> score), against the oracle's dense operator, and its dispatch rules.
reference: src/gp_model.py:441-452 (mu_Sigma_pred), the score of src/acquisition.py:72-81 at G = 1."""
import os

import numpy as np
import pytest

from oracle import ppbo_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(fused):
    """An Engine whose ctx read PPBO_FUSED = `fused` when it was created (the knob is per ctx, read once)."""
    from ppbo_amd.engine import Engine
    old = os.environ.get("PPBO_FUSED")
    os.environ["PPBO_FUSED"] = str(fused)
    try:
        return Engine(0)
    finally:
        if old is None:
            del os.environ["PPBO_FUSED"]
        else:
            os.environ["PPBO_FUSED"] = old


@pytest.fixture(scope="module")
def engines():
    e = {k: _engine(k) for k in (0, 1, 2)}
    yield e
    for v in e.values():
        v.close()


def host(t):
    return t.cpu().numpy()


def synth_post(eng, N, D, m, kernel, theta, seed=0):
    """A posterior state with random entries in the layout ppbo_posterior produces: Lambda in star form (no edge on
    observation rows), G block lower triangular with explicit zeros right of a row's star."""
    from ppbo_amd.engine import Posterior
    rng = np.random.default_rng(seed)
    mblk = m + 1
    assert N % mblk == 0
    X = rng.random((N, D))
    alpha = rng.standard_normal(N)
    lam_diag = -np.abs(rng.standard_normal(N)) * 0.3
    lam_off = np.abs(rng.standard_normal(N)) * 0.1
    lam_off[::mblk] = 0.0
    G = rng.standard_normal((N, N)) * 0.05
    kend = ((np.arange(N) // mblk) + 1) * mblk
    G[np.arange(N)[None, :] >= kend[:, None]] = 0.0
    arrs = dict(X=X, alpha=alpha, lam_diag=lam_diag, lam_off=lam_off, G=G)
    return Posterior(kernel, tuple(theta), m, eng.dev(X), eng.dev(alpha), eng.dev(lam_diag), eng.dev(lam_off), eng.dev(G)), arrs


def dense_reference(arrs, Xc, kernel, theta, m):
    """mu and sigma^2 from the same state in NumPy: sigma^2 = sigma_f^2 + k*' Lambda k* + |G k*|^2."""
    Ks = orc.KERNELS[kernel](arrs["X"], Xc, theta)            # [N, M], raw cross-covariance
    N, mblk = arrs["X"].shape[0], m + 1
    Lam = np.diag(arrs["lam_diag"])
    for j in range(N):
        if j % mblk:
            o = j - j % mblk
            Lam[j, o] = Lam[o, j] = arrs["lam_off"][j]
    mu = Ks.T @ arrs["alpha"]
    Y = arrs["G"] @ Ks
    var = theta[2] ** 2 + np.einsum("jc,jk,kc->c", Ks, Lam, Ks) + (Y * Y).sum(axis=0)
    return mu, var


SHAPES = [  # N, D, m, kernel, M: one pass / two passes, stars that divide nothing, D off every multiple of 4, M off 32
    (32, 1, 31, "SE_kernel", 1), (64, 2, 31, "SE_kernel", 33), (78, 4, 25, "SE_kernel", 100), (200, 5, 9, "RQ_kernel", 1000),
    (240, 16, 14, "SE_kernel", 257), (256, 6, 31, "SE_kernel", 4096), (260, 3, 25, "RQ_kernel", 31), (264, 7, 32, "SE_kernel", 500),
    (416, 6, 25, "SE_kernel", 3000), (462, 2, 32, "SE_kernel", 777), (512, 6, 31, "SE_kernel", 16384), (480, 13, 39, "RQ_kernel", 64),
]


@pytest.mark.parametrize("N,D,m,kernel,M", SHAPES)
def test_fused_matches_three_launch_and_dense(engines, N, D, m, kernel, M):
    th = (0.001, 0.26, 0.1) if kernel == "SE_kernel" else (0.3, 0.6, 0.8)
    e1, e0 = engines[1], engines[0]
    p1, arrs = synth_post(e1, N, D, m, kernel, th, seed=N + D)
    p0, _ = synth_post(e0, N, D, m, kernel, th, seed=N + D)
    Xc = np.random.default_rng(1).random((M, D))
    e1.profile(True)
    e0.profile(True)
    for kind in (0, 1, 2):        # PPBO_SCORE_MEAN (with variance requested), POINTWISE_EI, VARIANCE
        o1 = e1.predict(p1, Xc, score=kind, mustar=0.1, want_score=True)
        o0 = e0.predict(p0, Xc, score=kind, mustar=0.1, want_score=True)
        for k in ("mu", "var", "score"):
            a, b = host(o1[k]), host(o0[k])
            assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-300), (kind, k)
        sc = host(o1["score"])
        assert o1["best_idx"] == int(np.argmax(sc)) and o1["best_val"] == sc[o1["best_idx"]]
    assert e1.profile_read("fused_score")[1] == 3 and e1.profile_read("quadform")[1] == 0     # the one-launch path ran ...
    assert e0.profile_read("fused_score")[1] == 0 and e0.profile_read("quadform")[1] == 3     # ... and PPBO_FUSED=0 did not
    e1.profile(False)
    e0.profile(False)
    mu_ref, var_ref = dense_reference(arrs, Xc, kernel, th, m)
    o1 = e1.predict(p1, Xc, score=1, mustar=0.1)
    assert np.abs(host(o1["mu"]) - mu_ref).max() <= 1e-11 * np.abs(mu_ref).max()
    assert np.abs(host(o1["var"]) - var_ref).max() <= 1e-11 * np.abs(var_ref).max()
    # the same candidates in two calls of other lengths (other tile counts, another ragged last block): the same bits
    if M >= 3:
        k = M // 2 + 1
        a, b = e1.predict(p1, Xc[:k], want_best=False), e1.predict(p1, Xc[k:], want_best=False)
        assert np.array_equal(np.concatenate([host(a["var"]), host(b["var"])]), host(o1["var"]))
        assert np.array_equal(np.concatenate([host(a["mu"]), host(b["mu"])]), host(o1["mu"]))


@pytest.mark.parametrize("N,D,m,kernel,M", [(650, 2, 25, "SE_kernel", 3000), (1024, 10, 31, "SE_kernel", 4096),
                                            (512, 20, 31, "SE_kernel", 1000), (992, 24, 30, "RQ_kernel", 555)])
def test_sixteen_wavefront_form_is_opt_in_and_agrees(engines, N, D, m, kernel, M):
    """Up to ~1000 rows (or more than 16 dimensions) the kernel exists but measured slower than the three-launch form
    (one workgroup per CU: nothing hides its K* phases): PPBO_FUSED=2 selects it, the default does not."""
    th = (0.001, 0.26, 0.1) if kernel == "SE_kernel" else (0.3, 0.6, 0.8)
    out = {}
    for k in (1, 2):
        e = engines[k]
        p, arrs = synth_post(e, N, D, m, kernel, th, seed=7)
        Xc = np.random.default_rng(2).random((M, D))
        e.profile(True)
        out[k] = e.predict(p, Xc, score=1, mustar=0.05, want_score=True)
        out[k, "n"] = e.profile_read("fused_score")[1]
        e.profile(False)
    assert out[1, "n"] == 0 and out[2, "n"] == 1
    for k in ("mu", "var", "score"):
        a, b = host(out[2][k]), host(out[1][k])
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), k
    mu_ref, var_ref = dense_reference(arrs, Xc, kernel, th, m)
    assert np.abs(host(out[2]["var"]) - var_ref).max() <= 1e-11 * np.abs(var_ref).max()


def test_dispatch_rules(engines, golden):
    """What stays on the three-launch form by default: the camphor kernel, the fp32-K* report, mean-only scoring, models
    whose second pass does not fit one panel.  The choice never depends on the candidate count."""
    e = engines[1]
    e.profile(True)

    def launches(post, Xc, **kw):
        e.profile_reset()
        e.predict(post, Xc, **kw)
        return e.profile_read("fused_score")[1]

    rng = np.random.default_rng(3)
    p, _ = synth_post(e, 256, 6, 31, "camphor_copper_kernel", (0.001, 0.26, 0.1))
    assert launches(p, rng.random((100, 6)), score=1) == 0
    p, _ = synth_post(e, 256, 6, 31, "SE_kernel", (0.001, 0.26, 0.1))
    assert launches(p, rng.random((100, 6)), score=1, kstar_fp32=True) == 0
    assert launches(p, rng.random((100, 6)), score=0, want_var=False) == 0          # mean only: no G, no contraction
    for M in (1, 31, 32, 33, 5000, 70000):
        assert launches(p, rng.random((M, 6)), score=1) == 1
    p, _ = synth_post(e, 1040, 4, 25, "SE_kernel", (0.3, 0.6, 0.8))
    assert launches(p, rng.random((64, 4)), score=1) == 0
    e.profile(False)


def test_fused_on_a_fitted_model_vs_the_reference(engines, golden):
    """The reference's own mu / diag Sigma_pred (tests/golden/smoke.npz, N = 64) through both forms."""
    g = golden("smoke")
    X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
    for k in (0, 1):
        e = engines[k]
        Sinv = e.pd_inverse(e.gram(X, th, kern))
        post = e.posterior(X, th, kern, Sinv, g["fMAP"], m)
        out = e.predict(post, g["Xc"], score=1, mustar=float(np.max(g["mu"])))
        assert np.abs(host(out["mu"]) - g["mu"]).max() <= 1e-6 * np.abs(g["mu"]).max()
        assert np.abs(host(out["var"]) - g["var"]).max() <= 1e-6 * float(th[2]) ** 2
