"""Drop-in Hsampler (src/random_fourier_sampler.py): random-Fourier-feature posterior samples of
the utility and their maximisers, with the feature projection, the weight-space Laplace terms
and the candidate scoring evaluated by the HIP kernels (ppbo_rff_project / _terms / _score).

Differences in *cost*, not in results: the reference's weight-space Hessian is diagonal
(:118-122) but is returned, inverted and sampled as a dense F x F matrix (:134-140, 207-213);
here the diagonal is kept as a vector (the dense forms are still produced on request for
attribute compatibility).  The L-BFGS-B multi-start of return_xstar (:143-176) is one device
enqueue: batched candidate scoring, start selection and the whole multi-start gradient ascent (ppbo_rff_search).
"""
from __future__ import annotations

import time

import numpy as np

from .engine import get_engine

SCORE_CANDIDATES = 65536
RFF_STARTS = 32            # refined starts per posterior sample (the reference: 5-30 L-BFGS-B runs)


class Hsampler:
    def __init__(self, gp_model, nFeatures=1000, engine=None):
        self.eng = engine if engine is not None else getattr(gp_model, "eng", None) or get_engine()
        self.nFeatures = nFeatures
        self.b = None
        self.W = None
        self.D = gp_model.D
        self.m = gp_model.m
        self.X = gp_model.X
        self.GP_xstar = gp_model.xstar
        self.GP_xstars_local = gp_model.xstars_local
        self.n_gausshermite_sample_points = gp_model.n_gausshermite_sample_points
        self.obs_indices = gp_model.obs_indices
        self.kernel = str(gp_model.kernel.__name__)
        self.theta = gp_model.theta
        self._phi_X = None        # host copy of Phi(X), made on first access (67 MB at C3: 5 ms nobody needs per update)
        self.omega_MAP = None
        self.cov_diag = None      # the posterior covariance of omega is diagonal (S_hessian is): this is its diagonal
        self._hess_diag = None
        self.verbose = False
        self._dPhi = None
        # the model's resident uniform candidate pool (GPModel._candidate_pool: drawn once per model, rotated per use),
        # shared instead of drawing and uploading 65536 x D fresh uniforms per sampler (4 ms of a 13 ms cycle at D = 20)
        self._pool_of = getattr(gp_model, "_candidate_pool", None)
        # device copies of W, b and X, made once per ARRAY OBJECT (assigning a new array refreshes them; the basis is 655 KB
        # at F = 4096, D = 20, and every projection / search used to upload it again)
        self._dcache = {}
        dX = getattr(gp_model, "_dX", None)
        if dX is not None and tuple(dX.shape) == tuple(np.shape(self.X)) and dX.device == self.eng.device:
            self._dcache["X"] = (self.X, dX)

    def _dev(self, name):
        arr = getattr(self, name)
        hit = self._dcache.get(name)
        if hit is None or hit[0] is not arr:
            hit = (arr, self.eng.dev(np.asarray(arr, dtype=float).ravel() if name == "b" else arr))
            self._dcache[name] = hit
        return hit[1]

    # ---- basis -----------------------------------------------------------------
    def generate_basis(self):
        if self.kernel == "SE_kernel":                        # only the SE spectral density is supported (:40-42)
            self.W = np.random.randn(self.nFeatures, self.D) / self.theta[1]
        self.b = np.random.uniform(low=0, high=2 * np.pi, size=self.nFeatures)[:, None]

    def _scale(self):
        return np.sqrt(2.0 * self.theta[2] ** 2 / self.nFeatures)

    def phiVec(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        return self.eng.rff_project(x, self.W, self.b.ravel(), self.theta[2]).cpu().numpy()

    def phi(self, x):
        return self._scale() * np.cos(self.W @ np.asarray(x, dtype=float) + self.b.ravel())

    def Dphi(self, x):
        return -self._scale() * np.sin(self.W @ np.asarray(x, dtype=float) + self.b.ravel())[:, None] * self.W

    def DDphi(self, x):
        raise NotImplementedError

    def update_phi_X(self):
        self._dPhi = self.eng.rff_project(self._dev("X"), self._dev("W"), self._dev("b"), self.theta[2])
        self._phi_X = None

    @property
    def phi_X(self):
        """Phi(X) [F, N] as a NumPy array (random_fourier_sampler.py:57-58); lives on the device, copied on demand."""
        if self._phi_X is None and self._dPhi is not None:
            self._phi_X = self._dPhi.cpu().numpy()
        return self._phi_X

    @phi_X.setter
    def phi_X(self, value):
        self._phi_X = None if value is None else np.asarray(value, dtype=float)
        self._dPhi = None if value is None else self.eng.dev(self._phi_X)

    @property
    def covariance_inv(self):
        """-S_hessian(omega_MAP) as the dense F x F matrix the reference stores (:136); built on demand."""
        return None if self._hess_diag is None else np.diag(self._hess_diag)

    @property
    def covariance(self):
        """The dense F x F posterior covariance of the reference (:137); built on demand from its diagonal."""
        return None if self.cov_diag is None else np.diag(self.cov_diag)

    # ---- weight-space Laplace terms ------------------------------------------------
    def _terms(self, omega, theta):
        return self.eng.rff_terms(self._dPhi, omega, self.m, theta[0])

    def S(self, omega, theta):
        return self._terms(omega, theta)[0]

    def S_grad(self, omega, theta):
        return self._terms(omega, theta)[1].cpu().numpy()

    def S_hessian_diag(self, omega, theta):
        return self._terms(omega, theta)[2].cpu().numpy()

    def S_hessian(self, omega, theta):
        return np.diag(self.S_hessian_diag(omega, theta))

    # ---- the reference's per-query helpers of S (random_fourier_sampler.py:62-102), kept for callers that use them;
    # S / S_grad / S_hessian above do NOT go through them (ppbo_rff_terms sums over all queries in one pass)
    def sum_Phi(self, i, order_of_derivative, f, sigma, sample_points=None, weights=None):
        """Query i (an element of obs_indices): order 0 -> sum_j Phi(Delta_j / sqrt 2) (scalar, closed form of the
        Gauss-Hermite integral); order 1 -> sum_j (phi_X[:, i+1+j] - phi_X[:, i]) var2_normal_pdf(Delta_j) (a vector
        of nFeatures); order 2 -> sum_j -(phi_X[:, i+1+j] - phi_X[:, i])^2 Delta_j / 2 var2_normal_pdf(Delta_j).  The
        per-pseudo-observation weights come from ppbo_sum_phi's kernel family (ppbo_laplace_terms), the feature
        contractions from the device GEMM on the resident Phi(X)."""
        if order_of_derivative not in (0, 1, 2):
            print("The derivatives of an order higher than 2 are not needed!")
            return None
        i, m = int(i), self.m
        f = np.asarray(f, dtype=float).ravel()
        if order_of_derivative == 0:
            return float(self.eng.sum_phi(f, m, sigma, 0).cpu().numpy()[i // (m + 1)])
        # beta (order 1) and Lambda's off-diagonal (order 2) carry exactly these weights, up to their scale factors
        _, beta, _, lo = self.eng.laplace_terms(f, m, sigma)
        blk = slice(i + 1, i + m + 1)
        if order_of_derivative == 1:
            w = -(beta[blk] * (sigma * m))                    # var2_normal_pdf(Delta_j)
        else:
            w = lo[blk] * (m * sigma ** 2)                    # -Delta_j / 2 var2_normal_pdf(Delta_j)
        diff = self._dPhi[:, blk] - self._dPhi[:, i:i + 1]    # [F, m] on the device
        if order_of_derivative == 2:
            diff = diff * diff
        return self.eng.dgemm(diff.contiguous(), w.reshape(-1, 1).contiguous()).cpu().numpy().ravel()

    def sum_Phi_vec(self, order_of_derivative, f, sigma):
        """One sum_Phi per observation (random_fourier_sampler.py:96-102): [n_q] for order 0, [n_q, nFeatures] else."""
        out = [self.sum_Phi(i, order_of_derivative, f, sigma) for i in self.obs_indices]
        return None if any(o is None for o in out) else np.array(out)

    def update_omega_MAP(self):
        """Maximise S from a standard-normal start (:124-132).  The Hessian is diagonal, so the
        trust-region Newton of the reference reduces to per-coordinate safeguarded Newton steps."""
        omega = np.random.randn(self.nFeatures)
        start = time.time()
        # the whole trust-region loop runs behind one call (ppbo_rff_omega_map) and on the device: omega, gradient,
        # Hessian diagonal and the region's state stay there, S / |grad S| / the iteration count come back once
        omega, S, gnorm, iters = self.eng.rff_omega_map(self._dPhi, omega, self.m, self.theta[0], maxiter=500, gtol=1e-6)
        self.omega_MAP_stats = {"S": S, "gradnorm": gnorm, "iterations": iters}
        if self.verbose:
            print("... this took " + str(time.time() - start) + " seconds.")
        self.omega_MAP = omega

    def update_covariancematrix(self):
        hd = -self.S_hessian_diag(self.omega_MAP, self.theta)
        if np.any(hd <= 0):
            print("---!!!--- Posterior covariance matrix is not PSD ---!!!---")
            return
        self._hess_diag = hd
        self.cov_diag = 1.0 / hd

    def sample_omega(self):
        if self.cov_diag is None:
            print("Omega sampler error! Omega MAP-estimate was used instead.")
            return self.omega_MAP
        return self.omega_MAP + np.sqrt(self.cov_diag) * np.random.standard_normal(self.nFeatures)

    # ---- maximiser of one posterior sample --------------------------------------------
    def score_candidates(self, Xc, omega):
        """phi(x)^T omega for many candidates on the device; returns (scores, best value, best index)."""
        sc, bv, bi = self.eng.rff_score(Xc, self._dev("W"), self._dev("b"), self.theta[2], omega)
        return sc.cpu().numpy(), bv, bi

    def return_xstar(self, omega):
        """argmax_x phi(x)^T omega (:143-176).  The reference runs 5-30 L-BFGS-B searches from perturbed local maxima
        of the posterior mean on NumPy phi / Dphi; here ONE device enqueue (ppbo_rff_search) scores a rotated resident
        uniform pool plus such perturbations, keeps the RFF_STARTS best that are > 0.05 apart and runs the whole
        projected gradient ascent of each inside one kernel; the best refined point is returned."""
        import torch
        start = time.time()
        D = self.D
        pool = self.__dict__.get("_pool")
        if pool is None:
            shared = self._pool_of() if self._pool_of is not None else None
            if shared is not None and shared.shape[1] == D and shared.device == self.eng.device:
                pool = self._pool = shared
            else:
                pool = self._pool = self.eng.dev(np.random.uniform(0, 1, (SCORE_CANDIDATES, D)))
        M = pool.shape[0]
        loc = np.atleast_2d(self.GP_xstars_local)
        k = min(len(loc) * 64, SCORE_CANDIDATES // 4)
        near = np.clip(loc[np.random.randint(len(loc), size=k)] + 0.01 * np.random.uniform(0, 1, (k, D)), 0, 1)
        work = torch.empty((M + k, D), dtype=torch.float64, device=self.eng.device)
        self.eng.shift_points(pool, np.random.uniform(0, 1, D), out=work[:M])
        work[M:].copy_(self.eng.dev(near))
        # 100 Barzilai-Borwein iterations per start: the winning start is stationary after ~50 (tests/probes/
        # rff_ascent_scale.py: the same maximum at 50, 100 and 200), the cap only bounds the starts that keep bouncing
        xs, vals = self.eng.rff_search(work, self._dev("W"), self._dev("b"), self.theta[2], omega, K=RFF_STARTS, iters=100)
        if self.verbose:
            print("Optimization of f_approx took " + str(time.time() - start) + " seconds.")
        if len(vals) == 0 or not np.isfinite(vals).any():
            return None
        return xs[int(np.nanargmax(vals))]

    def return_xstar_for_dim(self, omega, dim, x_ref):
        x_ref = np.array(x_ref, dtype=float)
        grid = np.tile(x_ref, (4096, 1))
        grid[:, dim - 1] = np.linspace(0, 1, 4096)
        _, _, bi = self.eng.rff_score(grid, self._dev("W"), self._dev("b"), self.theta[2], omega, want_score=False)
        return grid[bi]

    def sample_xstar(self):
        xstar = None
        while xstar is None:
            xstar = self.return_xstar(self.sample_omega())
        return xstar

    def sample_xstar_for_dim(self, dim, x_ref):
        return self.return_xstar_for_dim(self.sample_omega(), dim, x_ref)
