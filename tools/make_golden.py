#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE ITSELF (build container only).

Imports /root/reference/src in memory with the four shims of SURVEY.md 8c
(stub arspy, stub GPyOpt, scipy.linalg.solve sym_pos->assume_a,
scipy.optimize.minimize x0.ravel()).  Nothing from the reference is copied:
the outputs are data (inputs + expected outputs, float64) written as small
.npz files under tests/golden/.  /root/reference does not exist on the GPU
box, so this script only ever runs here; tests load the committed fixtures.

usage: python tools/make_golden.py [smoke rq c2 c4 c3 c5]
"""
from __future__ import annotations

import os
import sys
import time
import types

import numpy as np
import scipy
import scipy.linalg
import scipy.optimize

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CONFIGS = {
    # name: D, n_q, theta, kernel, bounds, do_evidence, rff_F, rff_map
    "smoke": dict(D=3, n_q=2, theta=[0.09, 0.3, 0.5], kernel="SE_kernel", F=96, ev=True, omap=True),
    "rq": dict(D=4, n_q=4, theta=[0.09, 0.3, 0.5], kernel="RQ_kernel", F=0, ev=True, omap=False),
    "c2": dict(D=6, n_q=16, theta=[0.001, 0.26, 0.1], kernel="SE_kernel", F=1000, ev=False, omap=False),
    "c4": dict(D=10, n_q=32, theta=[0.001, 0.4, 0.15], kernel="SE_kernel", F=0, ev=False, omap=False),
    "c3": dict(D=20, n_q=64, theta=[0.09, 0.3, 0.5], kernel="SE_kernel", F=4096, ev=False, omap=False),
    "c5": dict(D=6, n_q=128, theta=[0.001, 0.26, 0.1], kernel="camphor_copper_kernel", F=0, ev=False, omap=False,
               bounds=((-0.5, 0.5), (-0.5, 0.5), (4, 7), (-180, 180), (-180, 180), (-180, 180))),
    "cam_small": dict(D=6, n_q=4, theta=[0.001, 0.26, 0.1], kernel="camphor_copper_kernel", F=0, ev=False, omap=False,
                      bounds=((-0.5, 0.5), (-0.5, 0.5), (4, 7), (-180, 180), (-180, 180), (-180, 180))),
}
M_PSEUDO = 31


def install_shims():
    sys.dont_write_bytecode = True
    ars = types.ModuleType("arspy")
    ars_ars = types.ModuleType("arspy.ars")

    def _absent(*a, **k):
        raise RuntimeError("arspy is not installed in the build container")

    ars_ars.adaptive_rejection_sampling = _absent
    ars.ars = ars_ars
    sys.modules["arspy"] = ars
    sys.modules["arspy.ars"] = ars_ars

    gpy = types.ModuleType("GPyOpt")
    gpm = types.ModuleType("GPyOpt.methods")

    class BayesianOptimization:  # noqa: D401
        def __init__(self, *a, **k):
            raise RuntimeError("GPyOpt is not installed in the build container")

    gpm.BayesianOptimization = BayesianOptimization
    gpy.methods = gpm
    sys.modules["GPyOpt"] = gpy
    sys.modules["GPyOpt.methods"] = gpm

    _solve = scipy.linalg.solve

    def solve(a, b, sym_pos=None, **kw):
        if sym_pos is not None:
            kw["assume_a"] = "pos" if sym_pos else "gen"
        return _solve(a, b, **kw)

    scipy.linalg.solve = solve
    _minimize = scipy.optimize.minimize

    def minimize(fun, x0, *a, **k):
        return _minimize(fun, np.asarray(x0).ravel(), *a, **k)

    scipy.optimize.minimize = minimize
    sys.path.insert(0, REF)


def build_design(gp_mod, settings_mod, cfg):
    """Golden recipe of SURVEY.md 8c: seed 0, xi = e_{q mod D}, x uniform with
    x[q mod D]=0, alpha uniform; 'equispaced' pseudo-observations, m=31."""
    D, n_q = cfg["D"], cfg["n_q"]
    bounds = cfg.get("bounds", ((0, 1),) * D)
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    np.random.seed(0)
    rows = []
    for q in range(n_q):
        d = q % D
        xi = np.zeros(D)
        xi[d] = 1.0
        x = lo + np.random.rand(D) * (hi - lo)
        x[d] = 0.0
        a = lo[d] + np.random.rand() * (hi[d] - lo[d])
        rows.append(np.concatenate([a * xi + x, xi, [a]]))
    X_obs = np.array(rows)
    st = settings_mod.PPBO_settings(D=D, bounds=bounds, xi_acquisition_function="PCD",
                                    theta_initial=list(cfg["theta"]), m=cfg.get("m", M_PSEUDO), verbose=False,
                                    kernel=cfg["kernel"])
    gp = gp_mod.GPModel(st)
    gp.update_feedback_processing_object(X_obs)
    gp.update_data()
    return gp, st, X_obs


def lam_to_compact(Lam, m):
    N = Lam.shape[0]
    diag = np.diag(Lam).copy()
    idx = np.arange(N)
    latest = (idx // (m + 1)) * (m + 1)
    off = Lam[latest, idx].copy()
    off[idx % (m + 1) == 0] = 0.0
    # structure check: nothing outside diag + star edges
    chk = Lam.copy()
    chk[idx, idx] = 0
    chk[latest, idx] = 0
    chk[idx, latest] = 0
    assert np.abs(chk).max() == 0.0
    return diag, off


def likelihood_only_terms(gp, fs):
    """The reference's T / T_grad evaluated with Sigma_inv_ = 0: the pure
    likelihood part and beta, free of the cond(Sigma) noise of Sigma^-1 f."""
    Z = np.zeros((gp.N, gp.N))
    return dict(lap_Tlik=np.array([float(gp.T(f, gp.theta, Z)) for f in fs]),
                lap_beta=np.stack([gp.T_grad(f, gp.theta, Z) for f in fs]))


def augment(name):
    """Add fields to an existing fixture without redoing the slow fit."""
    import gp_model as ref_gp
    import ppbo_settings as ref_settings
    path = os.path.join(OUT, f"{name}.npz")
    z = np.load(path)
    out = {k: z[k] for k in z.files}
    gp, _, _ = build_design(ref_gp, ref_settings, CONFIGS[name])
    assert np.array_equal(np.asarray(gp.X), out["X"])
    gp.set_theta()
    out.update(likelihood_only_terms(gp, out["lap_f"]))
    np.savez_compressed(path, **out)
    print(f"[{name}] augmented {path}")


def design_only(name):
    """Write just the reference-built design (X, X_obs) so a start vector can be prepared elsewhere."""
    import gp_model as ref_gp
    import ppbo_settings as ref_settings
    gp, _, X_obs = build_design(ref_gp, ref_settings, CONFIGS[name])
    path = os.path.join(OUT, f"_{name}_design.npz")
    np.savez_compressed(path, X=np.asarray(gp.X), X_obs=X_obs, theta=np.array(CONFIGS[name]["theta"], dtype=float),
                        m=gp.m, kernel=CONFIGS[name]["kernel"])
    print(f"[{name}] wrote {path}")


START_VECTOR = None   # optional near-optimal start for the reference's own trust-exact run (N=4096 takes hours cold)


def run_config(name):
    import gp_model as ref_gp
    import ppbo_settings as ref_settings
    import random_fourier_sampler as ref_rff
    import acquisition as ref_acq

    cfg = CONFIGS[name]
    t0 = time.time()
    gp, st, X_obs = build_design(ref_gp, ref_settings, cfg)
    N, D, m = gp.N, gp.D, gp.m
    out = dict(name=name, X=np.asarray(gp.X), X_obs=X_obs, theta=np.array(cfg["theta"], dtype=float),
               m=m, D=D, N=N, kernel=cfg["kernel"], obs_indices=np.array(gp.obs_indices),
               latest_obs_indices=np.array(gp.latest_obs_indices),
               bounds=np.array(cfg.get("bounds", ((0, 1),) * D), dtype=float))
    rng = np.random.default_rng(7)

    # ---- G1: Sigma --------------------------------------------------------
    gp.set_theta()
    gp.update_Sigma(gp.theta)
    Sig = gp.Sigma
    c = min(64, N)
    ii = rng.integers(0, N, 2048)
    jj = rng.integers(0, N, 2048)
    out.update(Sigma_corner=Sig[:c, :c].copy(), Sigma_rowsum=Sig.sum(axis=1), Sigma_trace=np.trace(Sig),
               Sigma_ii=ii, Sigma_jj=jj, Sigma_samples=Sig[ii, jj].copy())
    raw = gp.create_Gramian_nonsquare(gp.X, gp.X[:c], gp.kernel, gp.theta)
    out.update(Kraw_cols=raw.copy())          # unregularised K(X, X[:c])
    gp.update_Sigma_inv(gp.theta)
    print(f"[{name}] N={N} D={D} Sigma+inv {time.time()-t0:.1f}s", flush=True)

    # ---- G3: fMAP with a stored start (patched RNG draw) ------------------
    f_init = np.random.default_rng(2).multivariate_normal(np.zeros(N), Sig, method="cholesky")
    if START_VECTOR is not None:
        f_init = np.load(START_VECTOR).astype(np.float64).ravel()
        assert f_init.shape == (N,)
        out["f_init_is_warm_start"] = True
    _mvn = np.random.multivariate_normal
    np.random.multivariate_normal = lambda mean, cov, *a, **k: f_init.copy()
    try:
        gp.fMAP = None
        gp.update_fMAP()
    finally:
        np.random.multivariate_normal = _mvn
    fMAP = np.asarray(gp.fMAP).ravel()
    g_at_map = gp.T_grad(fMAP, gp.theta)
    alpha = gp.Sigma_inv.dot(fMAP)
    out.update(f_init=f_init, fMAP=fMAP, gradnorm_fMAP=np.linalg.norm(g_at_map), alpha=alpha,
               T_fMAP=float(gp.T(fMAP, gp.theta)))
    print(f"[{name}] fMAP |g|={np.linalg.norm(g_at_map):.3e} {time.time()-t0:.1f}s", flush=True)

    # ---- G2: Laplace terms at three f vectors ------------------------------
    fs = np.stack([f_init, np.zeros(N), fMAP])
    Ts, Gs, Ld, Lo = [], [], [], []
    for f in fs:
        Ts.append(float(gp.T(f, gp.theta)))
        Gs.append(gp.T_grad(f, gp.theta))
        dg, of = lam_to_compact(gp.create_Lambda(f, gp.theta[0]), m)
        Ld.append(dg)
        Lo.append(of)
    out.update(lap_f=fs, lap_T=np.array(Ts), lap_grad=np.stack(Gs), lap_diag=np.stack(Ld), lap_off=np.stack(Lo))
    out.update(likelihood_only_terms(gp, fs))

    # ---- posterior as in update_model :111-117 -----------------------------
    gp.Lambda_MAP = gp.create_Lambda(gp.fMAP, gp.theta[0])
    gp.posterior_covariance_inv = gp.Sigma_inv - gp.Lambda_MAP
    gp.posterior_covariance = ref_gp.pd_inverse(gp.posterior_covariance_inv)
    out.update(P_diag=np.diag(gp.posterior_covariance).copy(),
               P_corner=gp.posterior_covariance[:c, :c].copy())

    # ---- G4: candidates ----------------------------------------------------
    Mc = 512
    Xc = rng.random((Mc, D))
    near = gp.X[rng.integers(0, N, Mc // 2)] + 0.02 * rng.standard_normal((Mc // 2, D))
    Xc[Mc // 2:] = np.clip(near, 0, 1)
    mu, Spred = gp.mu_Sigma_pred(Xc)
    out.update(Xc=Xc, mu=np.asarray(mu).ravel(), var=np.diag(Spred).copy())
    mu1 = np.array([gp.mu_pred(x) for x in Xc[:16]])
    out.update(mu_pred16=mu1)
    print(f"[{name}] predict {time.time()-t0:.1f}s", flush=True)

    # ---- G8: one projective line: grid, mu, cov, reference EI/varmax -------
    rec = {}
    _msp = gp.mu_Sigma_pred

    def rec_msp(Xp):
        r = _msp(Xp)
        rec["grid"], rec["mu"], rec["cov"] = np.array(Xp), np.asarray(r[0]).ravel(), np.array(r[1])
        return r

    gp.mu_Sigma_pred = rec_msp
    gp.mustar = float(np.max(mu))
    d = 1 % D
    xi = np.zeros(D)
    xi[d] = 1.0
    xl = rng.random(D)
    xl[d] = 0.0
    np.random.seed(123)
    ei_ref = ref_acq.EI(xi, xl, gp, 150)
    np.random.seed(123)
    vm_ref = ref_acq.varmax(xi, xl, gp, 150)
    # many-sample reference values for a tighter statistical check
    np.random.seed(321)
    ei_big = ref_acq.EI(xi, xl, gp, 4000)
    gp.mu_Sigma_pred = _msp
    out.update(line_xi=xi, line_x=xl, line_grid=rec["grid"], line_mu=rec["mu"], line_cov=rec["cov"],
               line_mustar=gp.mustar, line_ei_ref150=ei_ref, line_varmax_ref150=vm_ref, line_ei_ref4000=ei_big)

    # ---- G5: evidence (small configs) --------------------------------------
    if cfg["ev"]:
        th_list = [list(cfg["theta"]), [1.0, 0.3, 0.5], [1.0, 0.15, 1.2]]
        ev_vals, ev_inits = [], []
        for k, th in enumerate(th_list):
            f0 = np.random.default_rng(20 + k).multivariate_normal(np.zeros(N), Sig, method="cholesky")
            np.random.multivariate_normal = lambda mean, cov, *a, f0=f0, **kw: f0.copy()
            try:
                ev_vals.append(float(gp.evidence(th, None)))
            finally:
                np.random.multivariate_normal = _mvn
            ev_inits.append(f0)
        out.update(ev_theta=np.array(th_list), ev_value=np.array(ev_vals), ev_finit=np.stack(ev_inits))
        print(f"[{name}] evidence {ev_vals} {time.time()-t0:.1f}s", flush=True)

    # ---- G6: RFF ------------------------------------------------------------
    if cfg["F"]:
        F = cfg["F"]
        gp.xstar = Xc[int(np.argmax(mu))]
        gp.xstars_local = gp.xstar.reshape(1, D)
        hs = ref_rff.Hsampler(gp, F)
        np.random.seed(3)
        hs.generate_basis()
        hs.update_phi_X()
        Phi = hs.phi_X
        omega = np.random.default_rng(5).standard_normal(F)
        S = float(hs.S(omega, hs.theta))
        Sg = hs.S_grad(omega, hs.theta)
        Sh = np.diag(hs.S_hessian(omega, hs.theta)).copy()
        fc = min(32, F)
        scores = np.array([float(np.dot(hs.phi(x).T, omega)) for x in Xc])
        out.update(rff_W=hs.W, rff_b=hs.b.ravel(), rff_Phi_corner=Phi[:fc, :c].copy(), rff_Phi_rowsum=Phi.sum(axis=1),
                   rff_Phi_colsum=Phi.sum(axis=0), rff_omega=omega, rff_S=S, rff_Sgrad=Sg, rff_Shdiag=Sh,
                   rff_scores=scores, rff_Dphi0=hs.Dphi(Xc[0]).T.dot(omega))
        if cfg["omap"]:
            om0 = np.random.default_rng(6).standard_normal(F)
            _randn = np.random.randn
            np.random.randn = lambda *a: om0.copy()
            try:
                hs.update_omega_MAP()
            finally:
                np.random.randn = _randn
            hs.update_covariancematrix()
            out.update(rff_omega0=om0, rff_omega_MAP=hs.omega_MAP, rff_cov_diag=np.diag(hs.covariance).copy())
        print(f"[{name}] rff {time.time()-t0:.1f}s", flush=True)

    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB) in {time.time()-t0:.1f}s", flush=True)


if __name__ == "__main__":
    install_shims()
    args = sys.argv[1:]
    if args and args[0] == "--augment":
        for nm in args[1:]:
            augment(nm)
    elif args and args[0] == "--design-only":
        for nm in args[1:]:
            design_only(nm)
    elif args and args[0] == "--start":
        START_VECTOR = args[1]
        for nm in args[2:]:
            run_config(nm)
    else:
        for nm in (args or ["smoke", "rq", "cam_small", "c2"]):
            run_config(nm)
