#!/usr/bin/env python3
"""Round-2 golden vectors, again produced by running the REFERENCE ITSELF (build container only; same four
in-memory shims as tools/make_golden.py, nothing copied).  They pin what round 1 only checked for shape:

  g7.npz        C1 trace: the six-hump-camel PCD run of ppbo_numerical_main.py:57-144 --
                4 corner initial queries + 21 PCD queries, m = 25, theta [0.01, 0.26, 0.1], seed 0 -- as the
                reference's own GPModel / next_query / pp_sixhump_camel produce it: X_obs rows, and per query N,
                f_MAP, xstar (scaled and unscaled), mustar, xstars_local count.
  <cfg>_x.npz   extras on an existing fixture's model state (X, theta, f_MAP taken from <cfg>.npz so the slow fit is
                not repeated):  mu_star outputs (gp_model.py:415-437), next_query for PCD / EXT / RAND /
                EI-EXT-FAST / EI-EXT with seeded RNG incl. the dispatcher's bookkeeping (acquisition.py:9-65),
                per-direction EI values, a 4000-draw varmax of the stored line (acquisition.py:170-178),
                Hsampler.return_xstar for the stored omega (random_fourier_sampler.py:143-176), evidence
                (gp_model.py:278-319) at three thetas.

Reproducibility: everything is seeded, but differential evolution (mu_star, the simulated user of g7) follows the
bits of its objective, and those depend on the BLAS summation order: the committed files were produced in the 8-core
build container with the DEFAULT OpenBLAS thread count (two runs are bit-identical there; with OPENBLAS_NUM_THREADS=2
c2_x's DE results move by 3e-3 in x*, 3e-6 in mu*).  The `augment-*` commands add keys to an existing file.

usage: python tools/make_golden_r2.py g7 | tgn | extras smoke rq c2 c4 c3 | augment-rff-dim smoke c2 | augment-omega-map c2
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (shims + design recipe)

OUT = mg.OUT


def g7():
    import gp_model as ref_gp
    import ppbo_settings as ref_settings
    import acquisition as ref_acq
    import misc as ref_misc
    sys.path.insert(0, "/root/reference/numerical_experiments")
    import test_functions as ref_tf

    t0 = time.time()
    m, n_actual, seed = 25, 21, 0
    theta = [0.01, 0.26, 0.1]
    bounds = ((-3, 3), (-2, 2))
    st = ref_settings.PPBO_settings(D=2, bounds=bounds, xi_acquisition_function="PCD", m=m, theta_initial=list(theta),
                                    alpha_grid_distribution="equispaced", verbose=False)
    np.random.seed(seed)                                        # ppbo_numerical_main.py:135
    xi0 = np.diag([st.original_bounds[i][1] for i in range(st.D)])
    xi0 = np.tile(xi0, (2, 1))                                  # :139
    x0 = ref_misc.hypercube_corners(st.original_bounds)[0:len(xi0)]
    D = st.D
    results = np.empty((0, 2 * D + 1))
    rec = dict(N=[], fMAP=[], xstar=[], xstar_scaled=[], mustar=[], n_local=[], init=[], gradnorm=[])
    gp = None

    def record(is_init):
        rec["N"].append(gp.N)
        rec["fMAP"].append(np.asarray(gp.fMAP).ravel().copy())
        rec["xstar_scaled"].append(np.asarray(gp.xstar).copy())
        rec["xstar"].append(np.asarray(gp.FP.unscale(gp.xstar)).ravel().copy())
        rec["mustar"].append(float(gp.mustar))
        rec["n_local"].append(int(np.asarray(gp.xstars_local).shape[0]))
        rec["init"].append(bool(is_init))
        rec["gradnorm"].append(float(np.linalg.norm(gp.T_grad(np.asarray(gp.fMAP).ravel(), gp.theta))))

    # ---- the loop of run_ppbo_loop (ppbo_numerical_main.py:57-127), re-driven because the file itself cannot be
    # imported (hard-coded wd_root :14, pypet); ADAPTIVE_INITIALIZATION = False, no theta optimisation (:186-190)
    for i in range(len(xi0)):
        if i == len(xi0) - 1:
            gp.turn_initialization_off()
        x = np.array(x0[i], dtype=float)
        xi = np.array(xi0[i], dtype=float)
        x[xi != 0] = 0
        a = ref_tf.pp_sixhump_camel(xi, x)
        results = np.vstack([results, list(a * xi + x) + list(xi) + [a]])
        if i == 0:
            gp = ref_gp.GPModel(st)
        gp.update_feedback_processing_object(np.array(results))
        gp.update_data()
        gp.update_model()
        record(i < len(xi0) - 1)
        print(f"[g7] init {i + 1}/{len(xi0)} N={gp.N} mustar={gp.mustar:.6f} {time.time() - t0:.0f}s", flush=True)
    gp.turn_initialization_off()
    nq_xi, nq_x, nq_dim = [], [], []
    for i in range(n_actual):
        if i + 1 == len(xi0) + n_actual:                        # :104 (never true with initial queries, as in the reference)
            gp.set_last_iteration()
        xi_next, x_next = ref_acq.next_query(st, gp, unscale=True)
        nq_xi.append(np.array(xi_next).copy())
        nq_x.append(np.array(x_next).copy())
        nq_dim.append(int(st.dim_query_prev_iter))
        a = ref_tf.pp_sixhump_camel(xi_next, x_next)
        results = np.vstack([results, list(a * xi_next + x_next) + list(xi_next) + [a]])
        gp.update_feedback_processing_object(np.array(results))
        gp.mustar_previous_iteration = gp.mustar
        gp.update_data()
        gp.update_model(optimize_theta=False)
        record(False)
        xs = rec["xstar"][-1]
        print(f"[g7] query {i + 1}/{n_actual} N={gp.N} xstar={xs} mustar={gp.mustar:.6f} {time.time() - t0:.0f}s", flush=True)

    nmax = max(rec["N"])
    fpad = np.full((len(rec["N"]), nmax), np.nan)
    for k, f in enumerate(rec["fMAP"]):
        fpad[k, :len(f)] = f
    out = dict(X_obs=results, m=m, D=D, theta=np.array(theta), bounds=np.array(bounds, dtype=float), seed=seed,
               n_init=len(xi0), n_actual=n_actual, N=np.array(rec["N"]), fMAP=fpad, X_final=np.asarray(gp.X),
               xstar=np.array(rec["xstar"]), xstar_scaled=np.array(rec["xstar_scaled"]), mustar=np.array(rec["mustar"]),
               n_local=np.array(rec["n_local"]), is_init=np.array(rec["init"]), gradnorm_fMAP=np.array(rec["gradnorm"]),
               next_xi=np.array(nq_xi), next_x=np.array(nq_x), next_dim=np.array(nq_dim),
               true_optimum=np.array([[0.0898, -0.7126], [-0.0898, 0.7126]]))
    path = os.path.join(OUT, "g7.npz")
    np.savez_compressed(path, **out)
    print(f"[g7] wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB) in {time.time() - t0:.0f}s")


def _model_from_fixture(name):
    """Reference GPModel in the state the fixture's update_model left it (f_MAP from the fixture, no refit)."""
    import gp_model as ref_gp
    import ppbo_settings as ref_settings
    z = dict(np.load(os.path.join(OUT, f"{name}.npz")))
    gp, st, _ = mg.build_design(ref_gp, ref_settings, mg.CONFIGS[name])
    assert np.array_equal(np.asarray(gp.X), z["X"])
    gp.set_theta()
    gp.update_Sigma(gp.theta)
    gp.update_Sigma_inv(gp.theta)
    gp.fMAP = z["fMAP"].copy()
    gp.Lambda_MAP = gp.create_Lambda(gp.fMAP, gp.theta[0])
    gp.posterior_covariance_inv = gp.Sigma_inv - gp.Lambda_MAP
    gp.posterior_covariance = ref_gp.pd_inverse(gp.posterior_covariance_inv)
    gp.initialization_running = False
    return gp, st, z, ref_gp, ref_settings


def extras(name, do_evidence):
    import acquisition as ref_acq
    import random_fourier_sampler as ref_rff
    t0 = time.time()
    gp, st, z, ref_gp, ref_settings = _model_from_fixture(name)
    D = gp.D
    out = dict(name=name)
    print(f"[{name}_x] model state rebuilt {time.time() - t0:.0f}s", flush=True)

    # ---- varmax / EI of the stored line with many draws (statistical anchors) ----------------------------------
    gp.mustar = float(z["line_mustar"])
    np.random.seed(322)
    out["line_varmax_ref4000"] = ref_acq.varmax(z["line_xi"], z["line_x"], gp, 4000)
    np.random.seed(323)
    out["line_varmax_ref4000_b"] = ref_acq.varmax(z["line_xi"], z["line_x"], gp, 4000)   # spread between two runs
    print(f"[{name}_x] varmax4000 {out['line_varmax_ref4000']:.6e} / {out['line_varmax_ref4000_b']:.6e} "
          f"{time.time() - t0:.0f}s", flush=True)

    # ---- mu_star (differential evolution, seeded) ---------------------------------------------------------------
    np.random.seed(40)
    xstar, mustar, xloc = gp.mu_star(mustar_finding_trials=3)
    out.update(mustar_seed=40, xstar=np.array(xstar), mustar=float(mustar), xstars_local=np.array(xloc).reshape(-1, D))
    mu_loc = np.array([gp.mu_pred(x) for x in np.array(xloc).reshape(-1, D)])
    out["mu_at_xstars_local"] = mu_loc
    gp.xstar, gp.mustar, gp.xstars_local = xstar.copy(), mustar, np.array(xloc).reshape(-1, D)
    print(f"[{name}_x] mu_star {mustar:.8f} at {np.round(xstar, 4)} ({len(mu_loc)} local) {time.time() - t0:.0f}s", flush=True)

    # ---- next_query dispatcher (acquisition.py:9-65) with seeded RNG -------------------------------------------
    def settings(acq, xacq="exploit"):
        s = ref_settings.PPBO_settings(D=D, bounds=tuple(map(tuple, z["bounds"])), xi_acquisition_function=acq,
                                       theta_initial=list(z["theta"]), m=int(z["m"]), verbose=False, kernel=str(z["kernel"]))
        s.x_acquisition_function = xacq
        return s

    for acq, xacq, ncalls in (("PCD", "exploit", D + 1), ("EXT", "exploit", D + 1), ("RAND", "exploit", 3),
                              ("RAND", "random", 3), ("PCD", "random", 2)):
        s = settings(acq, xacq)
        key = f"nq_{acq}_{xacq}".replace("-", "")
        xi_l, x_l, dim_l = [], [], []
        for k in range(ncalls):
            np.random.seed(500 + k)
            xi_u, x_u = ref_acq.next_query(s, gp, unscale=True)
            xi_l.append(np.array(xi_u))
            x_l.append(np.array(x_u))
            dim_l.append(int(getattr(s, "dim_query_prev_iter", -1)))
        out.update({key + "_xi": np.array(xi_l), key + "_x": np.array(x_l), key + "_dim": np.array(dim_l)})
    # EI-driven direction choice: record every EI the dispatcher evaluates
    ei_log = []
    _EI = ref_acq.EI

    def EI_rec(xi, x, GP_model, mc):
        v = _EI(xi, x, GP_model, mc)
        ei_log.append((np.array(xi).copy(), np.array(x).copy(), float(v)))
        return v

    ref_acq.EI = EI_rec
    try:
        for acq in ("EI-EXT-FAST", "EI-EXT"):
            if acq == "EI-EXT" and gp.N > 1100:
                continue                                 # D*50 reference EI calls at N = 2048 take an hour
            s = settings(acq, "exploit")
            del ei_log[:]
            np.random.seed(600)
            xi_u, x_u = ref_acq.next_query(s, gp, unscale=True)
            key = "nq_" + acq.replace("-", "")
            out.update({key + "_xi": np.array(xi_u), key + "_x": np.array(x_u),
                        key + "_ei_xi": np.array([e[0] for e in ei_log]), key + "_ei_x": np.array([e[1] for e in ei_log]),
                        key + "_ei_val": np.array([e[2] for e in ei_log])})
            print(f"[{name}_x] {acq}: xi={np.round(xi_u, 3)} from {len(ei_log)} EI calls {time.time() - t0:.0f}s", flush=True)
    finally:
        ref_acq.EI = _EI

    # ---- Hsampler.return_xstar for the stored omega -------------------------------------------------------------
    if "rff_W" in z:
        F = z["rff_W"].shape[0]
        hs = ref_rff.Hsampler(gp, F)
        hs.W, hs.b = z["rff_W"].copy(), z["rff_b"].reshape(F, 1).copy()
        hs.update_phi_X()
        np.random.seed(70)
        xs = hs.return_xstar(z["rff_omega"])
        out.update(rff_xstar=np.array(xs), rff_xstar_val=float(np.dot(hs.phi(xs).T, z["rff_omega"])), rff_xstar_seed=70)
        print(f"[{name}_x] return_xstar val {out['rff_xstar_val']:.8f} {time.time() - t0:.0f}s", flush=True)

    # ---- evidence (gp_model.py:278-319) ---------------------------------------------------------------------------
    if do_evidence:
        th_list = [list(map(float, z["theta"])), [1.0, 0.3, 0.5], [1.0, 0.15, 1.2]]
        vals, inits = [], []
        _mvn = np.random.multivariate_normal
        for k, th in enumerate(th_list):
            f0 = np.random.default_rng(20 + k).multivariate_normal(np.zeros(gp.N), gp.Sigma, method="cholesky")
            np.random.multivariate_normal = lambda mean, cov, *a, f0=f0, **kw: f0.copy()
            try:
                vals.append(float(gp.evidence(th, None)))
            finally:
                np.random.multivariate_normal = _mvn
            inits.append(f0)
            print(f"[{name}_x] evidence{th} = {vals[-1]:.8f} {time.time() - t0:.0f}s", flush=True)
        out.update(ev_theta=np.array(th_list), ev_value=np.array(vals), ev_finit=np.stack(inits))

    path = os.path.join(OUT, f"{name}_x.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}_x] wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB) in {time.time() - t0:.0f}s", flush=True)


def augment_rff_dim(name):
    """Add Hsampler.return_xstar_for_dim outputs (random_fourier_sampler.py:180-204, Nelder-Mead per coordinate from
    GP_xstar) to an existing <name>_x.npz without touching what is already in it."""
    import random_fourier_sampler as ref_rff
    gp, st, z, ref_gp, ref_settings = _model_from_fixture(name)
    path = os.path.join(OUT, f"{name}_x.npz")
    x = dict(np.load(path))
    gp.xstar, gp.mustar, gp.xstars_local = x["xstar"].copy(), float(x["mustar"]), x["xstars_local"].copy()
    F = z["rff_W"].shape[0]
    hs = ref_rff.Hsampler(gp, F)
    hs.W, hs.b = z["rff_W"].copy(), z["rff_b"].reshape(F, 1).copy()
    hs.update_phi_X()
    om = z["rff_omega"]
    pts, vals = [], []
    for dim in range(1, gp.D + 1):
        xr = hs.return_xstar_for_dim(om, dim, x["xstar"].copy())
        pts.append(np.array(xr))
        vals.append(float(np.dot(hs.phi(xr).T, om)))
    x.update(rff_xstar_dim=np.array(pts), rff_xstar_dim_val=np.array(vals))
    np.savez_compressed(path, **x)
    print(f"[{name}_x] return_xstar_for_dim values {np.round(vals, 6)}")


def augment_omega_map(name):
    """Add Hsampler.update_omega_MAP / update_covariancematrix outputs (random_fourier_sampler.py:124-140) at the fixture's
    F (c2: F = 1000; round 1 only had F = 96) to <name>_x.npz; the start vector replaces the reference's randn draw."""
    import random_fourier_sampler as ref_rff
    gp, st, z, ref_gp, ref_settings = _model_from_fixture(name)
    path = os.path.join(OUT, f"{name}_x.npz")
    x = dict(np.load(path))
    gp.xstar, gp.mustar, gp.xstars_local = x["xstar"].copy(), float(x["mustar"]), x["xstars_local"].copy()
    F = z["rff_W"].shape[0]
    hs = ref_rff.Hsampler(gp, F)
    hs.W, hs.b = z["rff_W"].copy(), z["rff_b"].reshape(F, 1).copy()
    hs.update_phi_X()
    om0 = np.random.default_rng(6).standard_normal(F)
    _randn = np.random.randn
    np.random.randn = lambda *a: om0.copy()
    t0 = time.time()
    try:
        hs.update_omega_MAP()
    finally:
        np.random.randn = _randn
    hs.update_covariancematrix()
    x.update(rff_omega0=om0, rff_omega_MAP=hs.omega_MAP, rff_cov_diag=np.diag(hs.covariance).copy(),
             rff_S_at_MAP=float(hs.S(hs.omega_MAP, hs.theta)))
    np.savez_compressed(path, **x)
    print(f"[{name}_x] omega_MAP F={F}: S = {x['rff_S_at_MAP']:.8f} in {time.time() - t0:.0f}s")


def tgn():
    """The reference's truncated-generalised-normal log-density (src/TGN_distribution.py:21-25) on grids: the
    sampler itself is arspy's adaptive rejection sampling (absent here), the DENSITY it samples is pinned by these."""
    import TGN_distribution as ref_tgn        # imports with the arspy stub; only log_TGN_pdf is used
    cases = [(5.0, 0.3, 0.0, 1.0), (2.0, -1.2, -3.0, 3.0), (2.6, 0.95, 0.0, 1.0), (3.5, 4.0, 4.0, 7.0), (2.05, 10.0, -180.0, 180.0)]
    xs, lp = [], []
    for gam, al, a, b in cases:
        x = np.linspace(a, b, 401)
        xs.append(x)
        lp.append(np.array([float(ref_tgn.log_TGN_pdf(v, gam, al, a, b)) for v in x]))
    path = os.path.join(OUT, "tgn.npz")
    np.savez_compressed(path, cases=np.array(cases), x=np.stack(xs), logpdf=np.stack(lp))
    print(f"[tgn] wrote {path}")


if __name__ == "__main__":
    mg.install_shims()
    args = sys.argv[1:]
    if not args:
        sys.exit(__doc__)
    if args[0] == "g7":
        g7()
    elif args[0] == "tgn":
        tgn()
    elif args[0] == "augment-omega-map":
        for nm in args[1:]:
            augment_omega_map(nm)
    elif args[0] == "augment-rff-dim":
        for nm in args[1:]:
            augment_rff_dim(nm)
    elif args[0] == "extras":
        for nm in args[1:]:
            extras(nm, do_evidence=(nm in ("c2",)))
