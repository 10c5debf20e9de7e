"""CPU statement (NumPy, uses the oracle for the likelihood terms) of the device algorithm behind
ppbo_fit_fmap_whitened: L-BFGS on phi(z) = -T(L z) in the prior-whitened variable z = L^-1 f
(Sigma = L L^T), where the Hessian I - L^T Lambda L has no trace of cond(Sigma) ~ 1e7.

    python tests/probes/whitened_lbfgs_proto.py c3,c2 [gtol]

Reference being matched: GPModel.update_fMAP (src/gp_model.py:354-389), same optimum, different path.
Not part of the product; the HIP kernels (csrc/fit.hip, lbfgs_step_kernel) implement exactly this recurrence
(vector-free two-loop on the Gram matrix of {s_i, y_i, g}, Armijo / approximate-Wolfe acceptance, cautious
pair updates, stop on the reference's own criterion |grad_f T| < gtol)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg as sl

from oracle import ppbo_oracle as orc

H = 8


def two_loop_vector_free(B, order):
    """coefficients delta over the basis [s_0..s_{H-1}, y_0..y_{H-1}, g] of d = -H_k g; order = ring slots,
    oldest first."""
    nb = 2 * H + 1
    delta = np.zeros(nb)
    delta[2 * H] = -1.0
    al = {}
    for r in reversed(order):
        a = (delta @ B[:, r]) / B[r, H + r]
        al[r] = a
        delta[H + r] -= a
    if order:
        r = order[-1]
        delta *= B[r, H + r] / B[H + r, H + r]
    for r in order:
        b = (delta @ B[:, H + r]) / B[r, H + r]
        delta[r] += al[r] - b
    return delta


def fit(name, gtol=1e-4, max_evals=3000, verbose=False):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "golden", f"{name}.npz")))
    X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
    sigma = th[0]
    S = orc.gram(X, th, kern)
    Sinv = orc.pd_inverse(S)
    L = np.linalg.cholesky(S)
    N = X.shape[0]
    gate = gtol * np.sqrt((L * L).sum())

    def evaluate(zt, need_gf):
        f = L @ zt
        lik = orc.sum_phi0(f, m, sigma).sum() / m
        beta = orc.beta_vector(f, m, sigma)
        gf2 = float(np.sum((Sinv @ f - beta) ** 2)) if need_gf else -1.0
        return 0.5 * zt @ zt + lik, zt - L.T @ beta, gf2

    basis = np.zeros((2 * H + 1, N))
    B = np.zeros((2 * H + 1, 2 * H + 1))
    z = np.zeros(N)
    zt = L.T @ (Sinv @ g["f_init"])
    hist = head = 0
    first, need_gf = True, False
    phi = dphi = alpha = 0.0
    evals = iters = ls = stall = 0
    gzbest = np.inf
    d = np.zeros(N)
    status = 0
    c1, c2, eps_f = 1e-4, 0.9, 1e-13
    while status == 0:
        phi_t, g_t, gf2 = evaluate(zt, need_gf)
        evals += 1
        gcur = basis[2 * H]
        if first:
            accept = True
        else:
            dphi_t = g_t @ d
            accept = phi_t <= phi + c1 * alpha * dphi or (
                phi_t <= phi + eps_f * max(1.0, abs(phi)) and (2 * c1 - 1) * dphi >= dphi_t >= c2 * dphi)
        if not np.isfinite(phi_t):
            accept = False
        if not accept:
            ls += 1
            if ls > 12:
                if hist > 0:
                    hist, ls = 0, 0
                    d = -gcur
                    dphi = -B[2 * H, 2 * H]
                    alpha = min(1.0, 1.0 / np.sqrt(-dphi))
                    zt = z + alpha * d
                    continue
                status = 3
                break
            an = -dphi * alpha * alpha / (2 * (phi_t - phi - dphi * alpha)) if np.isfinite(phi_t) else 0.0
            alpha = min(max(an, 0.1 * alpha), 0.5 * alpha)
            zt = z + alpha * d
            if evals >= max_evals:
                status = 5
            continue
        if not first:
            s, y = zt - z, g_t - gcur
            sy, ss, yy = s @ y, s @ s, y @ y
            if sy > 1e-10 * np.sqrt(ss * yy):
                r = head
                basis[r], basis[H + r] = s, y
                head = (head + 1) % H
                hist = min(hist + 1, H)
                basis[2 * H] = g_t
                for t in (r, H + r, 2 * H):
                    B[t, :] = basis @ basis[t]
                    B[:, t] = B[t, :]
            else:
                basis[2 * H] = g_t
                B[2 * H, :] = basis @ g_t
                B[:, 2 * H] = B[2 * H, :]
            gtgt = float(g_t @ g_t)
            if gtgt < 0.25 * gzbest:            # the gradient still shrinks by factors: not stagnation
                gzbest, stall = gtgt, 0
            else:
                stall = stall + 1 if phi - phi_t <= 1e-14 * max(1.0, abs(phi)) else 0
        else:
            basis[2 * H] = g_t
            B[2 * H, 2 * H] = g_t @ g_t
        z, phi = zt.copy(), phi_t
        iters += 0 if first else 1
        ls = 0
        gz = np.sqrt(B[2 * H, 2 * H])
        if verbose:
            print(f"  it {iters} evals {evals} phi {phi:.12f} |gz| {gz:.3e} |gf| {np.sqrt(gf2) if gf2 >= 0 else -1:.3e} a {alpha:.3g}")
        if gf2 >= 0 and gf2 < gtol * gtol:
            status = 1
            break
        if stall >= 5:
            status = 2
            break
        if evals >= max_evals:
            status = 5
            break
        need_gf = gz < gate
        order = [(head - hist + k) % H for k in range(hist)]
        delta = two_loop_vector_free(B, order)
        dphi = float(delta @ B[:, 2 * H])
        if not dphi < 0:
            hist = 0
            delta = np.zeros(2 * H + 1)
            delta[2 * H] = -1.0
            dphi = -B[2 * H, 2 * H]
        d = delta @ basis
        alpha = 1.0                     # also for the first move (was min(1, 1 / |g|): 19 -> 17 evaluations at c3, 70 -> 62 at c2)
        first = False
        zt = z + alpha * d
    f = L @ z
    gf = Sinv @ f - orc.beta_vector(f, m, sigma)
    print(f"{name}: status {status} iters {iters} evals {evals} phi {phi:.12f} (ref {-float(g['T_fMAP']):.12f}) "
          f"|gf| {np.linalg.norm(gf):.2e} (ref's {float(g['gradnorm_fMAP']):.2e}) max|f-fref| {np.abs(f - g['fMAP']).max():.2e}")
    return f


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["smoke"]
    gt = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
    for n in names:
        fit(n, gt, verbose=len(sys.argv) > 3)
