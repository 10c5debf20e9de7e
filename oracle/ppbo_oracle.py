"""CPU oracle for the PPBO GP-surrogate / acquisition hot path.

TEST INFRASTRUCTURE ONLY.  This module is a fresh NumPy/SciPy restatement of the
reference algorithm (AaltoPML/PPBO, pure Python).  It is the checker the HIP
path is compared with and the `cpu_baseline` leg of bench.py.  Nothing under
`ppbo_amd/` may import it; only `tests/`, `__graft_entry__.smoke()` and
`bench.py --cpu-baseline` do.

Pinning: every function here is checked in tests/test_oracle_golden.py against
golden vectors produced by running the *reference itself* in the build
container (tools/make_golden.py imports /root/reference/src with four
in-memory shims) and committed under tests/golden/.
Parity status: PINNED for kernels, regularisation, Laplace terms, f_MAP,
alpha, posterior mean/variance, line covariance, evidence, RFF features and
RFF Laplace terms.  UNPINNED: TGN grids (arspy absent) and GPyOpt outer
searches (GPyOpt absent) -- neither is on the hot path.

All citations are file:line in /root/reference.
Two cost modes exist where the reference does avoidable work:
  faithful=True   same operations/association order as the reference
  faithful=False  algebraically identical closed forms (used as "optimised CPU")
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import scipy.optimize
import scipy.stats
from scipy.special import ndtr

SHRINKAGE = 1e-6  # src/gp_model.py:26  COVARIANCE_SHRINKAGE
KERNEL_IDS = {"SE_kernel": 0, "RQ_kernel": 1, "camphor_copper_kernel": 2}


# --------------------------------------------------------------------------
# a-1 / a-2  kernels                                   src/kernels.py:3-53
# --------------------------------------------------------------------------
def sqdist(X1, X2):
    """Squared Euclidean distance by the expansion formula, clipped at 0
    (src/kernels.py:3-11)."""
    n1 = np.einsum("ij,ij->i", X1, X1)
    n2 = np.einsum("ij,ij->i", X2, X2)
    r2 = (n1[:, None] + n2[None, :]) - 2.0 * (X1 @ X2.T)
    np.maximum(r2, 0.0, out=r2)
    return r2


def sqdist_direct(X1, X2):
    """Direct-difference squared distance (what the HIP kernels evaluate).
    Differs from `sqdist` only by fp64 rounding (~1e-15 absolute)."""
    out = np.zeros((X1.shape[0], X2.shape[0]))
    for k in range(X1.shape[1]):
        dk = X1[:, k][:, None] - X2[:, k][None, :]
        out += dk * dk
    return out


def se_kernel(X1, X2, theta):
    """sigma_f^2 exp(-r^2 / (2 l^2))   (src/kernels.py:19-25)."""
    l, sf = theta[1], theta[2]
    return sf ** 2 * np.exp(-0.5 * sqdist(X1, X2) / l ** 2)


def rq_kernel(X1, X2, theta):
    """Rational quadratic with alpha fixed to 2 (src/kernels.py:27-34)."""
    l, sf = theta[1], theta[2]
    a = 2
    return sf ** 2 * (1.0 + sqdist(X1, X2) / (2 * a * l ** 2)) ** (-a)


def camphor_copper_kernel(X1, X2, theta):
    """Five period-1 periodic factors (dims 0,1,3,4,5) times an RBF on dim 2
    whose lengthscale is l+0.05 (src/kernels.py:36-53)."""
    l, sf = theta[1], theta[2]
    out = np.full((X1.shape[0], X2.shape[0]), sf ** 2)
    for k in (0, 1, 3, 4, 5):
        ad = np.abs(X1[:, k][:, None] - X2[:, k][None, :])
        out = out * np.exp(-2.0 * np.sin(np.pi * ad) ** 2 / l ** 2)
    dz = np.abs(X1[:, 2][:, None] - X2[:, 2][None, :])
    out = out * np.exp(-0.5 * dz ** 2 / (l + 0.05) ** 2)
    return out


KERNELS = {
    "SE_kernel": se_kernel,
    "RQ_kernel": rq_kernel,
    "camphor_copper_kernel": camphor_copper_kernel,
}


# --------------------------------------------------------------------------
# a-3  regularisation                                   src/misc.py:71-88
# --------------------------------------------------------------------------
def regularize_covariance(K, reg_level=SHRINKAGE, faithful=False, jitter=1e-7):
    """Negative diagonal -> jitter, [SVD round trip], shrink toward (tr K/n) I.
    The SVD round trip (src/misc.py:79-80) is the identity to ~1e-15 and is
    only executed when faithful=True (it is 3.9 s of the reference's 4.66 s
    update_Sigma at N=2048).  Shrink = sklearn.covariance.shrunk_covariance
    (src/misc.py:85): (1-s) K + s mu I, mu = tr(K)/n."""
    K = np.array(K, dtype=np.float64, copy=True)
    dg = np.diag(K).copy()
    dg[dg < 0] = jitter
    np.fill_diagonal(K, dg)
    if faithful:
        u, s, vh = np.linalg.svd(K, full_matrices=False)
        K = (u * s) @ vh
    n = K.shape[0]
    mu = np.trace(K) / n
    K = (1.0 - reg_level) * K
    K.flat[:: n + 1] += reg_level * mu
    return K


def gram(X, theta, kernel="SE_kernel", faithful=False):
    """create_Gramian(X,X)  (src/gp_model.py:147-151)."""
    return regularize_covariance(KERNELS[kernel](X, X, theta), SHRINKAGE, faithful)


def cross_cov(X, Xc, theta, kernel="SE_kernel"):
    """create_Gramian_nonsquare: raw kernel, no shrink (src/gp_model.py:153-155)."""
    return KERNELS[kernel](X, Xc, theta)


# --------------------------------------------------------------------------
# a-5  SPD inverse                                      src/misc.py:96-100
# --------------------------------------------------------------------------
def pd_inverse(M):
    """solve(M, I, sym_pos=True) == LAPACK posv (src/misc.py:96-100)."""
    return scipy.linalg.solve(M, np.eye(M.shape[0]), assume_a="pos", overwrite_b=True)


# --------------------------------------------------------------------------
# a-0  design bookkeeping              src/feedback_processing.py:162-165
# --------------------------------------------------------------------------
def design_indices(N, m):
    """Row 0 of every (m+1)-block is the observation, rows 1..m its pseudo
    observations.  Returns obs_indices, pseudobs_indices, latest_obs_indices."""
    idx = np.arange(N)
    obs = idx[idx % (m + 1) == 0]
    pse = idx[idx % (m + 1) != 0]
    latest = (idx // (m + 1)) * (m + 1)
    return obs, pse, latest


# --------------------------------------------------------------------------
# a-6 / a-7  Laplace terms                       src/gp_model.py:176-274
# --------------------------------------------------------------------------
_INV_SQRT_4PI = 1.0 / math.sqrt(4.0 * math.pi)


def var2_normal_pdf(x):
    """N(0,2) density (src/misc.py:134-135)."""
    return _INV_SQRT_4PI * np.exp(-0.25 * np.square(x))


def _deltas(f, m, sigma):
    """Delta[i, j] = (f[i(m+1)+1+j] - f[i(m+1)]) / sigma  (src/gp_model.py:185-186)."""
    F = np.asarray(f, dtype=np.float64).reshape(-1, m + 1)
    return (F[:, 1:] - F[:, :1]) / sigma


def sum_phi0(f, m, sigma, n_gh=None):
    """Per-query sum_j Phi(Delta_j / sqrt 2).  With n_gh set, the reference's
    Gauss-Hermite quadrature (src/gp_model.py:192, nodes :212); otherwise the
    closed form it converges to."""
    D = _deltas(f, m, sigma)
    if n_gh is None:
        return ndtr(D / math.sqrt(2.0)).sum(axis=1)
    pts, w = np.polynomial.hermite.hermgauss(n_gh)
    vals = ndtr(D[:, :, None] - math.sqrt(2.0) * pts[None, None, :]) @ w
    return vals.sum(axis=1) / math.sqrt(math.pi)


def T_value(f, Sigma_inv, m, sigma, n_gh=None):
    """T = -1/2 f' Sigma^-1 f - (1/m) sum sum Phi   (src/gp_model.py:221-226)."""
    f = np.asarray(f, dtype=np.float64).ravel()
    return -0.5 * f @ (Sigma_inv @ f) - sum_phi0(f, m, sigma, n_gh).sum() / m


def beta_vector(f, m, sigma):
    """beta of T_grad (src/gp_model.py:234-238): obs rows get
    sum_j phi2(Delta_j)/(sigma m), pseudo rows get -phi2(Delta_j)/(sigma m)."""
    D = _deltas(f, m, sigma)
    P = var2_normal_pdf(D) / (sigma * m)
    B = np.empty((D.shape[0], m + 1))
    B[:, 0] = P.sum(axis=1)
    B[:, 1:] = -P
    return B.ravel()


def T_grad(f, Sigma_inv, m, sigma):
    """-Sigma^-1 f + beta   (src/gp_model.py:228-240)."""
    f = np.asarray(f, dtype=np.float64).ravel()
    return -(Sigma_inv @ f) + beta_vector(f, m, sigma)


def lambda_compact(f, m, sigma):
    """Lambda of create_Lambda (src/gp_model.py:249-274) in star-graph form.
    Returns (diag[N], off[N]): off[j] is the (obs(j), j) off-diagonal weight
    for a pseudo row j and 0 on obs rows.
      pseudo diag  = +c/2 Delta phi2(Delta)            (:262)
      obs diag     = +c/2 sum_j Delta_j phi2(Delta_j)  (:258, sum_Phi order 2 is -1/2 sum ...)
      off (i, j)   = -c/2 Delta phi2(Delta)            (:271)
    with c = 1/(m sigma^2)."""
    D = _deltas(f, m, sigma)
    c = 1.0 / (m * sigma ** 2)
    w = 0.5 * c * D * var2_normal_pdf(D)
    diag = np.empty((D.shape[0], m + 1))
    off = np.zeros((D.shape[0], m + 1))
    diag[:, 0] = w.sum(axis=1)
    diag[:, 1:] = w
    off[:, 1:] = -w
    return diag.ravel(), off.ravel()


def lambda_dense(f, m, sigma):
    """Dense N x N Lambda exactly as the reference materialises it."""
    diag, off = lambda_compact(f, m, sigma)
    N = diag.shape[0]
    L = np.diag(diag)
    _, pse, latest = design_indices(N, m)
    L[latest[pse], pse] = off[pse]
    L[pse, latest[pse]] = off[pse]
    return L


def T_hessian(f, Sigma_inv, m, sigma):
    """-Sigma^-1 + Lambda   (src/gp_model.py:242-247)."""
    return -Sigma_inv + lambda_dense(f, m, sigma)


# --------------------------------------------------------------------------
# a-8  f_MAP                                     src/gp_model.py:354-389
# --------------------------------------------------------------------------
def fit_fmap_trust_exact(f_init, Sigma_inv, m, sigma, gtol=None, maxiter=None, n_gh=None):
    """The reference's optimiser call (src/gp_model.py:382-384): SciPy
    trust-exact on -T with exact gradient and dense Hessian."""
    opts = {"disp": False}
    if gtol is not None:
        opts["gtol"] = gtol
    if maxiter is not None:
        opts["maxiter"] = maxiter
    res = scipy.optimize.minimize(
        lambda f: -T_value(f, Sigma_inv, m, sigma, n_gh),
        np.asarray(f_init, dtype=np.float64).ravel(),
        method="trust-exact",
        jac=lambda f: -T_grad(f, Sigma_inv, m, sigma),
        hess=lambda f: -T_hessian(f, Sigma_inv, m, sigma),
        options=opts,
    )
    return res.x, res


def _tr_subproblem(H, g, radius, lam_prev_lb, shrink, k_easy=0.1, max_inner=60):
    """Nearly-exact trust-region step: min g'p + 1/2 p'Hp, |p| <= radius, by the
    More-Sorensen iteration on lam >= 0 with Cholesky of H + lam I (Conn, Gould
    & Toint, Trust-Region Methods, Alg. 7.3.4; Nocedal & Wright Alg. 4.3).
    The hard case is not refined (safeguarded bracketing only)."""
    n = g.size
    gnorm = np.linalg.norm(g)
    dg = np.diag(H)
    rs = np.abs(H).sum(axis=1) - np.abs(dg)
    hn = min(np.abs(H).sum(axis=1).max(), np.linalg.norm(H))
    lb = max(0.0, -dg.min(), gnorm / radius - min((dg + rs).max(), hn))
    ub = max(0.0, gnorm / radius + min(-(dg - rs).min(), hn))
    if shrink:
        lb = max(lb, lam_prev_lb)
    lam = 0.0 if lb == 0.0 else max(math.sqrt(lb * ub), lb + 0.01 * (ub - lb))
    nchol = 0
    p = np.zeros(n)
    boundary = True
    for _ in range(max_inner):
        nchol += 1
        try:
            c = scipy.linalg.cholesky(H + lam * np.eye(n), lower=True)
        except np.linalg.LinAlgError:
            lb = max(lb, lam)
            lam = max(math.sqrt(lb * ub), lb + 0.01 * (ub - lb)) if ub > lb else 2.0 * lam + 1e-12
            if ub <= lb:
                ub = 2.0 * lam
            continue
        w = scipy.linalg.solve_triangular(c, -g, lower=True)
        p = scipy.linalg.solve_triangular(c, w, lower=True, trans="T")
        pn = np.linalg.norm(p)
        if pn <= radius and lam == 0.0:
            boundary = False
            break
        if abs(pn - radius) <= k_easy * radius:
            break
        q = scipy.linalg.solve_triangular(c, p, lower=True)
        lam_new = lam + (pn / np.linalg.norm(q)) ** 2 * (pn - radius) / radius
        if pn < radius:
            ub = lam
        else:
            lb = lam
        if not (lb < lam_new < ub):
            lam_new = max(math.sqrt(lb * ub), lb + 0.01 * (ub - lb))
        if lam_new <= 0.0:
            lam_new = 0.0
        lam = lam_new
    return p, boundary, lb, nchol


def fit_fmap_newton(f_init, Sigma_inv, m, sigma, gtol=1e-6, maxiter=2000, stats=None):
    """Trust-region Newton on -T with the radius rules of SciPy's trust-region
    driver (initial radius 1, max 1000, eta 0.15, shrink x1/4 when rho<1/4, double
    when rho>3/4 on the boundary) and a More-Sorensen subproblem solve.  This is
    the CPU statement of the iteration that ppbo_fit (HIP) runs; the algorithm
    of record for parity is fit_fmap_trust_exact (SciPy, as the reference calls it)."""
    f = np.asarray(f_init, dtype=np.float64).ravel().copy()
    radius, rmax, eta = 1.0, 1000.0, 0.15
    phi = -T_value(f, Sigma_inv, m, sigma)
    g = -T_grad(f, Sigma_inv, m, sigma)
    H = Sigma_inv - lambda_dense(f, m, sigma)
    lam_lb, shrink, nchol = 0.0, False, 0
    it = 0
    while it < maxiter and np.linalg.norm(g) >= gtol:
        it += 1
        p, boundary, lam_lb, nc = _tr_subproblem(H, g, radius, lam_lb, shrink)
        nchol += nc
        pred = -(g @ p + 0.5 * p @ (H @ p))
        if pred <= 0:
            break
        fn = f + p
        phin = -T_value(fn, Sigma_inv, m, sigma)
        rho = (phi - phin) / pred
        old_radius = radius
        if rho < 0.25:
            radius *= 0.25
        elif rho > 0.75 and boundary:
            radius = min(2.0 * radius, rmax)
        shrink = radius < old_radius
        if rho > eta:
            f, phi = fn, phin
            g = -T_grad(f, Sigma_inv, m, sigma)
            H = Sigma_inv - lambda_dense(f, m, sigma)
        if radius < 1e-14:
            break
    if stats is not None:
        stats.update(iters=it, nchol=nchol, gradnorm=float(np.linalg.norm(g)))
    return f, it


# --------------------------------------------------------------------------
# posterior / prediction                  src/gp_model.py:111-117, 441-461
# --------------------------------------------------------------------------
def posterior_covariance(Sigma_inv, f_map, m, sigma):
    """P = (Sigma^-1 - Lambda_MAP)^-1   (src/gp_model.py:111,116-117)."""
    return pd_inverse(Sigma_inv - lambda_dense(f_map, m, sigma))


def variance_operator(Sigma_inv, P, faithful=True, lam=None):
    """A = Sigma^-1 - Sigma^-1 P Sigma^-1 (src/gp_model.py:449).  The
    non-faithful form is the identical operator written as W - W P W with
    W = -Lambda_MAP (Woodbury), which never touches Sigma^-1's 1e7 dynamic range."""
    if faithful:
        return Sigma_inv - Sigma_inv @ P @ Sigma_inv
    W = -lam
    return W - W @ P @ W


def mu_pred(x, X, theta, Sigma_inv, f_map, kernel="SE_kernel"):
    """(k' Sigma^-1) f_MAP for one point  (src/gp_model.py:454-458)."""
    k = cross_cov(X, np.asarray(x, dtype=np.float64).reshape(1, -1), theta, kernel)
    return float((k.T @ Sigma_inv @ f_map).item())


def mu_sigma_pred(Xc, X, theta, Sigma_inv, f_map, P, kernel="SE_kernel", faithful=True, A=None):
    """mu_Sigma_pred (src/gp_model.py:441-452).  faithful: (k'Sigma^-1) f
    association, regularised M x M prior block (SVD round trip), A rebuilt."""
    k = cross_cov(X, Xc, theta, kernel)
    if faithful:
        mu = k.T @ Sigma_inv @ f_map
        A = variance_operator(Sigma_inv, P, True)
    else:
        mu = k.T @ (Sigma_inv @ f_map)
    prior = regularize_covariance(KERNELS[kernel](Xc, Xc, theta), SHRINKAGE, faithful)
    return mu, prior - k.T @ A @ k


def predict_mean_var(Xc, X, theta, alpha, A, kernel="SE_kernel"):
    """Optimised-CPU candidate scoring: mu = K*' alpha, var = sigma_f^2 - diag(K*' A K*)
    (diag of src/gp_model.py:450; the shrunk prior has diagonal sigma_f^2 exactly)."""
    k = cross_cov(X, Xc, theta, kernel)
    mu = k.T @ alpha
    var = theta[2] ** 2 - np.einsum("ij,ij->j", k, A @ k)
    return mu, var


def mu_star(X, theta, Sigma_inv, f_map, kernel="SE_kernel", trials=3):
    """mu_star (src/gp_model.py:415-437): SciPy differential evolution (updating='immediate', maxiter=2000,
    global NumPy stream) on -mu_pred, `trials` times; distinct maxima (> 0.1 apart, :430) are collected.
    Returns xstar[D], mustar, xstars_local[k, D]."""
    D = X.shape[1]
    neg = lambda x: -mu_pred(x, X, theta, Sigma_inv, f_map, kernel)     # noqa: E731
    xstar = xloc = None
    best = np.inf
    for i in range(trials):
        res = scipy.optimize.differential_evolution(neg, ((0, 1),) * D, updating="immediate", disp=False, maxiter=2000)
        if i == 0:
            xstar, best, xloc = res.x, res.fun, res.x.reshape(1, D)
        else:
            if all(np.linalg.norm(x - res.x) > 1e-1 for x in xloc):
                xloc = np.vstack([xloc, res.x])
            if res.fun < best:
                best, xstar = res.fun, res.x
    return xstar.reshape(D), mu_pred(xstar, X, theta, Sigma_inv, f_map, kernel), xloc


def mean_grad(Xc, X, theta, alpha, kernel="SE_kernel"):
    """mu = K*' alpha and its gradient with respect to the (scaled) point.  The reference has no
    gradient function (mu_star maximises mu_pred by differential evolution, src/gp_model.py:415-437);
    this is the derivative of the kernels of src/kernels.py:19-53, pinned in the tests by central
    differences of the pinned mu."""
    Xc = np.atleast_2d(np.asarray(Xc, dtype=np.float64))
    l, sf = theta[1], theta[2]
    K = KERNELS[kernel](Xc, X, theta)                       # [M, N]
    W = K * alpha[None, :]
    mu = W.sum(axis=1)
    diff = Xc[:, None, :] - X[None, :, :]                   # [M, N, D]
    if kernel == "SE_kernel":
        g = -np.einsum("mn,mnd->md", W, diff) / l ** 2
    elif kernel == "RQ_kernel":
        r2 = np.einsum("mnd,mnd->mn", diff, diff)
        g = -np.einsum("mn,mnd->md", W / (1.0 + r2 / (4.0 * l ** 2)), diff) / l ** 2
    else:
        fac = -(2.0 * np.pi / l ** 2) * np.sin(2.0 * np.pi * diff)
        fac[:, :, 2] = -diff[:, :, 2] / (l + 0.05) ** 2
        g = np.einsum("mn,mnd->md", W, fac)
    return mu, g


def pointwise_ei(mu, var, mustar):
    """G=1 closed form of the EI Monte Carlo (src/acquisition.py:72-81):
    E[max(f - mustar, 0)], f ~ N(mu, var)."""
    s = np.sqrt(np.maximum(var, 0.0))
    d = mu - mustar
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(s > 0, d / s, 0.0)
    ei = d * ndtr(z) + s * scipy.stats.norm.pdf(z)
    return np.where(s > 0, ei, np.maximum(d, 0.0))


# --------------------------------------------------------------------------
# a-9  evidence                                  src/gp_model.py:278-319
# --------------------------------------------------------------------------
def log_prior(theta):
    """Log-normal hyper-priors (src/gp_model.py:287-290)."""
    p0 = scipy.stats.lognorm.pdf(theta[0], s=1, scale=np.exp(1))
    p1 = scipy.stats.lognorm.pdf(theta[1], s=0.5, scale=np.exp(-1.4))
    p2 = scipy.stats.lognorm.pdf(theta[2], s=0.5, scale=np.exp(1.7))
    return np.log(p0) + np.log(p1) + np.log(p2)


def evidence(theta, X, m, f_initial, kernel="SE_kernel", n_gh=None):
    """Laplace log-marginal likelihood + log-prior with the reference's quirks:
    M = I + Sigma Lambda (plus sign, :302) and sign*logdet summed over the LU
    factors (:307-310).  f_initial is explicit here (the reference draws it
    from N(0, self.Sigma), :294)."""
    Sig = gram(X, theta, kernel)
    Sinv = pd_inverse(Sig)
    f_map, _ = fit_fmap_trust_exact(f_initial, Sinv, m, theta[0], maxiter=500, n_gh=n_gh)
    Lam = lambda_dense(f_map, m, theta[0])
    Mtx = np.eye(X.shape[0]) + Sig @ Lam
    Pm, L, U = scipy.linalg.lu(Mtx)
    acc = 0.0
    for F in (Pm, L, U):
        s, ld = np.linalg.slogdet(F)
        acc += s * ld
    val = T_value(f_map, Sinv, m, theta[0], n_gh) - 0.5 * acc + log_prior(theta)
    if np.isnan(val) or not np.isfinite(val):
        return -500.0
    return float(val)


# --------------------------------------------------------------------------
# a-14  line EI / varmax                  src/acquisition.py:72-81,170-178
# --------------------------------------------------------------------------
def line_grid(xi, x, alphas):
    """xi_grid(...) for a scaled query: rows alpha_g * xi + x
    (src/feedback_processing.py:97-107) for a *given* alpha vector."""
    return np.outer(alphas, xi) + np.asarray(x)[None, :]


def line_samples(mu, cov, z, jitter=0.0):
    """f = mu + chol(cov) z for stored standard-normal z[S,G].  The reference
    draws through NumPy's SVD-based multivariate_normal (src/acquisition.py:79)
    which is not reproducible sample-for-sample; equality is in distribution."""
    G = mu.shape[0]
    C = 0.5 * (cov + cov.T) + jitter * np.eye(G)
    L = np.linalg.cholesky(C)
    return mu[None, :] + z @ L.T


def line_ei(mu, cov, z, mustar, jitter=0.0):
    fmax = line_samples(mu, cov, z, jitter).max(axis=1)
    return float(np.maximum(fmax - mustar, 0.0).mean())


def line_varmax(mu, cov, z, jitter=0.0):
    fmax = line_samples(mu, cov, z, jitter).max(axis=1)
    return float(np.mean((fmax - fmax.mean()) ** 2))


# --------------------------------------------------------------------------
# a-16 .. a-18  random Fourier features   src/random_fourier_sampler.py
# --------------------------------------------------------------------------
def rff_features(Xq, W, b, sigma_f):
    """phiVec: sqrt(2 sigma_f^2 / F) cos(W Xq' + b) -> [F, n]
    (src/random_fourier_sampler.py:45-47)."""
    F = W.shape[0]
    return math.sqrt(2.0 * sigma_f ** 2 / F) * np.cos(W @ Xq.T + b.reshape(-1, 1))


def rff_score(Xc, W, b, sigma_f, omega):
    """phi(x)' omega batched over candidates (src/random_fourier_sampler.py:166,170)."""
    return rff_features(Xc, W, b, sigma_f).T @ omega


def rff_return_xstar(W, b, sigma_f, omega, xstars_local, min_trials=5, max_trials=30):
    """Hsampler.return_xstar (src/random_fourier_sampler.py:143-176): L-BFGS-B on -phi(x)' omega from perturbed
    posterior-mean maxima (global NumPy stream), best in-bounds result of at least `min_trials` starts."""
    F, D = W.shape
    c = math.sqrt(2.0 * sigma_f ** 2 / F)
    phi = lambda x: c * np.cos(W @ x + b)                                   # noqa: E731  (:48-50)
    dphi = lambda x: -c * (np.sin(W @ x + b)[:, None] * W)                  # noqa: E731  (:51-53)
    fval, xstar, i = -1e10, None, 0
    loc = np.atleast_2d(xstars_local)
    while xstar is None or i < min_trials:
        if i > max_trials:
            break
        i += 1
        x0 = loc[np.random.randint(loc.shape[0])]
        x0 = np.clip(x0 + 0.01 * np.random.uniform(0, 1, size=D), 0, 1)
        res = scipy.optimize.minimize(lambda x: -float(phi(x) @ omega), x0=x0, method="L-BFGS-B", bounds=((0, 1),) * D,
                                      jac=lambda x: -(dphi(x).T @ omega), options={"disp": False, "maxiter": 5000})
        f = float(phi(res.x) @ omega)
        if f > fval and np.all((res.x >= 0) & (res.x <= 1)):
            fval, xstar = f, res.x
    return xstar, fval


def rff_terms(Phi, omega, m, sigma):
    """S, S_grad, diag(S_hessian)  (src/random_fourier_sampler.py:106-122).
    f = Phi' omega; column differences dPhi_ij = Phi[:, i+1+j] - Phi[:, i]."""
    F, N = Phi.shape
    f = Phi.T @ omega
    D = _deltas(f, m, sigma)                       # [n_q, m]
    S = -0.5 * omega @ omega - ndtr(D / math.sqrt(2.0)).sum() / m
    P3 = Phi.reshape(F, -1, m + 1)
    dPhi = P3[:, :, 1:] - P3[:, :, :1]             # [F, n_q, m]
    p2 = var2_normal_pdf(D)
    g = -omega - np.einsum("fqj,qj->f", dPhi, p2) / (sigma * m)
    h = -1.0 - np.einsum("fqj,qj->f", dPhi ** 2, -0.5 * D * p2) / (m * sigma ** 2)
    return float(S), g, h


def rff_omega_map(Phi, omega0, m, sigma):
    """update_omega_MAP (src/random_fourier_sampler.py:124-132): trust-exact
    with the reference's diagonal Hessian returned dense."""
    res = scipy.optimize.minimize(
        lambda w: -rff_terms(Phi, w, m, sigma)[0],
        np.asarray(omega0, dtype=np.float64).ravel(),
        method="trust-exact",
        jac=lambda w: -rff_terms(Phi, w, m, sigma)[1],
        hess=lambda w: -np.diag(rff_terms(Phi, w, m, sigma)[2]),
        options={"disp": False},
    )
    return res.x


# --------------------------------------------------------------------------
# synthetic workloads (SURVEY.md 8c/8d recipes) -- shared by tests and bench
# --------------------------------------------------------------------------
def synthetic_design(n_q, D, m=31, seed=0, noise=True):
    """Seeded design of the golden recipe: query q uses xi = e_{q mod D}, x
    uniform with x[q mod D] = 0, alpha* uniform; pseudo-observations on the
    'equispaced' noisy grid (src/feedback_processing.py:66-74) drawn from the
    same global stream order as the reference (alpha grid after each row).
    NOTE: this re-creates the *recipe*; the committed golden X comes from the
    reference's own FeedbackProcessing and is what parity tests load."""
    rs = np.random.RandomState(seed)
    rows = []
    for q in range(n_q):
        d = q % D
        x = rs.rand(D)
        x[d] = 0.0
        a = rs.rand()
        rows.append((d, x, a))
    X = np.empty((n_q * (m + 1), D))
    for q, (d, x, a) in enumerate(rows):
        xi = np.zeros(D)
        xi[d] = 1.0
        eps_b, eps_n = 0.005, 0.01
        while True:
            al = np.linspace(eps_b, 1 - eps_b, m)
            if noise:
                al = al + rs.normal(0, eps_n, m)
            al = np.unique(np.clip(al, 0, 1))
            if al.size == m:
                break
        blk = X[q * (m + 1):(q + 1) * (m + 1)]
        blk[0] = a * xi + x
        blk[1:] = np.outer(al, xi) + x[None, :]
    return X
