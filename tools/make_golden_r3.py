#!/usr/bin/env python3
"""Round-3 golden vectors, produced by running the REFERENCE ITSELF (build container only; the in-memory shims of
tools/make_golden.py, nothing copied).  They pin the small operators that round 3 exposes under the reference's own
names:

  compat_<cfg>.npz   on the model state of <cfg>.npz (X, theta, f_MAP, f_init):
      GPModel.sum_Phi_vec(order, f, sigma[, over_all_indices])  for order 0, 1, 2 at f_MAP and at f_init
      (gp_model.py:206-218, with the reference's Gauss-Hermite-200 quadrature for order 0), GPModel.sum_Phi(i, ...)
      at three observation rows (:176-204);
      misc.regularize_covariance (misc.py:71-88, SVD round trip and sklearn shrinkage included) on a symmetric test
      matrix with three negative diagonal entries, for (reg_level, pos_diag) in {(1e-4, True), (0.1, True),
      (0.05, False)}; misc.pd_inverse (:96-100) of the regularised matrix before the diagonal was spoiled;
      misc.is_positive_definite (:120-126) of that matrix (True) and of the spoiled one (False);
      Hsampler.sum_Phi_vec(order, f, sigma) (random_fourier_sampler.py:62-102) for order 0, 1, 2 on the fixture's
      basis (rff_W, rff_b) at f = Phi(X)^T rff_omega.

usage: python tools/make_golden_r3.py compat smoke rq
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402
import make_golden_r2 as r2  # noqa: E402

OUT = mg.OUT


def compat(name):
    import misc as ref_misc
    gp, st, z, ref_gp, _ = r2._model_from_fixture(name)
    sig = float(gp.theta[0])
    out = {}
    for tag, f in (("map", z["fMAP"]), ("init", z["f_init"])):
        f = np.asarray(f, dtype=float).ravel()
        for order in (0, 1, 2):
            out[f"sum_phi_{tag}_{order}"] = np.asarray(gp.sum_Phi_vec(order, f, sig), dtype=float)
        out[f"sum_phi_all_{tag}_1"] = np.asarray(gp.sum_Phi_vec(1, f, sig, over_all_indices=True), dtype=float)
    pts, w = np.polynomial.hermite.hermgauss(gp.n_gausshermite_sample_points)
    rows = [int(gp.obs_indices[k]) for k in (0, len(gp.obs_indices) // 2, len(gp.obs_indices) - 1)]
    out["sum_phi_rows"] = np.array(rows)
    out["sum_phi_scalar"] = np.array([[float(gp.sum_Phi(i, o, np.asarray(z["fMAP"]).ravel(), sig, pts, w)) for o in (0, 1, 2)]
                                      for i in rows])
    rng = np.random.default_rng(11)
    n = 40
    B = rng.standard_normal((n, n))
    K = B @ B.T / n + 0.3 * np.eye(n)
    out["pd_in"] = K.copy()                    # positive definite: the pd_inverse / is_positive_definite case
    K[3, 3], K[17, 17], K[39, 39] = -0.2, -1e-3, -4.0
    out["reg_in"] = K.copy()
    for k, (lev, pos) in enumerate(((1e-4, True), (0.1, True), (0.05, False))):
        out[f"reg_out_{k}"] = np.asarray(ref_misc.regularize_covariance(K.copy(), lev, pos))
        out[f"reg_arg_{k}"] = np.array([lev, float(pos)])
    out["pd_reg"] = np.asarray(ref_misc.regularize_covariance(out["pd_in"].copy(), 1e-4))
    out["pd_inv"] = np.asarray(ref_misc.pd_inverse(out["pd_reg"].copy()))
    out["is_pd"] = np.array([ref_misc.is_positive_definite(out["pd_reg"]), ref_misc.is_positive_definite(K)])
    # Hsampler.sum_Phi_vec (random_fourier_sampler.py:62-102) on the fixture's basis at f = Phi(X)^T omega
    if "rff_W" in z:                               # SE fixtures only: the reference has no other spectral basis
        import random_fourier_sampler as ref_rff
        F = z["rff_W"].shape[0]
        gp.xstar, gp.xstars_local = np.full(gp.D, 0.5), np.full((1, gp.D), 0.5)
        hs = ref_rff.Hsampler(gp, F)
        hs.W, hs.b = z["rff_W"].copy(), z["rff_b"].reshape(F, 1).copy()
        hs.update_phi_X()
        fw = np.asarray(hs.phi_X.T @ z["rff_omega"]).ravel()
        out["hs_f"] = fw
        for order in (0, 1, 2):
            out[f"hs_sum_phi_{order}"] = np.asarray(hs.sum_Phi_vec(order, fw, sig), dtype=float)
    path = os.path.join(OUT, f"compat_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[compat_{name}] wrote {path} ({os.path.getsize(path) / 1e3:.0f} kB); is_pd = {out['is_pd']}")


if __name__ == "__main__":
    mg.install_shims()
    args = sys.argv[1:]
    if not args:
        sys.exit(__doc__)
    if args[0] == "compat":
        for nm in args[1:]:
            compat(nm)
