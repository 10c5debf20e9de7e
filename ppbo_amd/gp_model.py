"""Drop-in GPModel: the reference's object surface (src/gp_model.py) on top of libppbo_hip.so.

Same constructor, methods, argument meaning and print-and-continue error behaviour as the
reference class; every O(N^2)/O(N^3)/O(M N^2) operation is a HIP kernel behind the C-ABI:
  update_Sigma       -> ppbo_gram            (gp_model.py:157-158, kernels.py, misc.py:71-88)
  update_Sigma_inv   -> ppbo_pd_inverse      (gp_model.py:161-162, misc.py:96-100)
  update_fMAP        -> ppbo_fit_fmap        (gp_model.py:354-389)
  Lambda / posterior -> ppbo_posterior       (gp_model.py:111-117, 249-274)
  mu_pred / mu_Sigma_pred -> ppbo_predict / ppbo_predict_cov   (gp_model.py:441-461)
  mu_star            -> batched candidate search with on-device argmax (replaces the sequential
                        differential evolution of gp_model.py:415-437), then a bounded polish.
Large matrices live on the GPU; the NumPy attributes other code reads (Sigma, Sigma_inv,
Lambda_MAP, posterior_covariance, ...) are materialised lazily on first access.
There is no CPU fallback: without the HIP library / a GPU the constructor raises.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.optimize

from . import kernels as _kernels
from .engine import NotPositiveDefinite, SCORE_MEAN, get_engine
from .feedback_processing import FeedbackProcessing

SEARCH_CANDIDATES = 65536      # uniform candidates per mu_star trial
ASCENT_STARTS = 32             # best well-separated candidates refined together on the device
ASCENT_ITERS = 100             # cap on ascent iterations per start (all inside one mean_ascent_kernel launch)
POLISH_GRAD_TOL = 1e-6         # |projected grad mu| / |mu| above which mu_star's winner gets the L-BFGS-B polish
APPEND_REFRESH = 64            # incremental mode: bordered updates of Sigma^-1 between two full inversions (measured:
                               # 32 appends leave |Sigma Sigma^-1 - I| where a full inversion leaves it, 8.2e-8 at N = 2048)


class GPModel:
    def __init__(self, PPBO_settings, engine=None, incremental=False, fMAP_method=None):
        """incremental=True (not a reference feature): when update_data() has only appended one
        query's m+1 rows (feedback_processing.py:133-154), Sigma^-1 is bordered instead of refactorised
        (ppbo_pd_inverse_append) and f_MAP starts from the previous estimate padded with the previous posterior
        mean at the new rows (the reference pads with the mean of f_MAP, gp_model.py:375-377, and only does so
        during initialisation; its default is a fresh prior draw per update)."""
        self.COVARIANCE_SHRINKAGE = 1e-6            # gp_model.py:26
        self.eng = engine if engine is not None else get_engine()
        self.incremental = bool(incremental)
        self._sinv_state = None      # (X copy, theta tuple, appends since the last full inversion) behind _dSigma_inv
        self.n_appends = 0           # bordered updates of Sigma^-1 actually performed (ppbo_pd_inverse_append)
        self.n_full_inversions = 0   # full potrf + trtri + GEMM inversions
        self.fMAP_restart_on_stall = False   # opt-in: refit from a fresh prior draw when the trust region stalls
        # mu_star ranks its 65536 candidates per trial by a posterior mean whose kernel values are evaluated in fp32
        # (ppbo_mean_search_multi, screen_fp32): the ranking only picks the ascents' starts, every reported value comes
        # from the fp64 ascent.  False: ranked by the fp64 mean (0.3 ms more per trial at C3).
        self.mustar_screen_fp32 = True
        self.polish_log = {"device_reascents": 0, "scipy_polishes": 0}    # how often mu_star's winners needed more than the ascent
        # "whitened": L-BFGS in z = L^-1 f finished by the trust region (ppbo_fit_fmap_whitened; O(N^2) per iteration; the
        # reference's optimum wherever T has one -- at sigma << sigma_f T has several local maxima and every local method,
        # SciPy's included, picks its own: DESIGN 5); anything else: the exact Newton trust region on f alone, which
        # follows SciPy trust-exact's iteration rules
        self.fMAP_method = fMAP_method if fMAP_method is not None else getattr(PPBO_settings, "fMAP_method", "whitened")
        if self.fMAP_method not in ("whitened", "trust-region"):
            raise ValueError("fMAP_method must be 'whitened' or 'trust-region'")
        self.fit_log = []            # one dict per update_fMAP trial: iterations, n_cholesky, warm, seconds
        s = PPBO_settings
        self.verbose = s.verbose
        self.FP = None
        self.D = s.D
        self.original_bounds = s.original_bounds
        self.bounds = ((0, 1),) * self.D
        self.X = None
        self.N = None
        self.m = s.n_pseudoobservations
        self.obs_indices = self.pseudobs_indices = self.latest_obs_indices = None
        self.alpha_grid_distribution = s.alpha_grid_distribution
        self.TGN_speed = s.TGN_speed
        self.n_gausshermite_sample_points = s.n_gausshermite_sample_points
        self.xi_acquisition_function = s.xi_acquisition_function
        self.kernel = _kernels.BY_NAME[s.kernel]    # the reference eval()s the string (gp_model.py:48)
        self.theta_initial = s.theta_initial
        self.theta = None
        self.fMAP = None
        self.fMAP_finding_trials = 1
        self.fMAP_optimizer = s.fMAP_optimizer
        self.fMAP_random_initial_vector = True
        self.fMAP_gtol = 1e-4                       # SciPy trust-exact default, which the reference inherits (gp_model.py:382-384)
        self.mustar_finding_trials = s.mustar_finding_trials
        self.mustar_previous_iteration = 0
        self.mustar = None
        self.xstar = None
        self.xstars_local = None
        self.initialization_running = True
        self.last_iteration = False
        self.skip_computations_during_initialization = s.skip_computations_during_initialization
        self.skip_xstaroptimization_during_initialization = s.skip_xstaroptimization_during_initialization
        self.fit_stats = None
        # device state
        self._dX = self._dSigma = self._dSigma_inv = self._dLinv = self._dL = None
        self._post = None           # engine.Posterior with G (variance operator)
        self._post_mean = None      # Posterior usable for the mean only (alpha current, no G)
        self._host = {}

    # ------------------------------------------------------------------ data
    def update_feedback_processing_object(self, X_obs):
        if self.FP is None:
            self.FP = FeedbackProcessing(self.D, self.m, self.original_bounds, self.alpha_grid_distribution,
                                         self.TGN_speed)
            self.FP.initialize_data(X_obs)
        else:
            self.FP.update_data(X_obs)

    def update_data(self):
        self.X = self.FP.X
        self.N = self.FP.N
        self.obs_indices = self.FP.obs_indices
        self.pseudobs_indices = self.FP.pseudobs_indices
        self.latest_obs_indices = self.FP.latest_obs_indices
        self._dX = self.eng.dev(self.X)

    def turn_initialization_off(self):
        self.initialization_running = False
        self.FP.alpha_grid_distribution = self.alpha_grid_distribution

    def set_last_iteration(self):
        self.last_iteration = True

    def is_pseudobs(self, i):
        return self.FP.is_pseudobs(i)

    # ------------------------------------------------------------------ lazily downloaded matrices
    def _lazy(self, key, make):
        if key not in self._host:
            self._host[key] = make()
        return self._host[key]

    def _invalidate(self, *keys):
        for k in keys:
            self._host.pop(k, None)

    @property
    def Sigma(self):
        return None if self._dSigma is None else self._lazy("Sigma", lambda: self._dSigma.cpu().numpy())

    @property
    def Sigma_inv(self):
        return None if self._dSigma_inv is None else self._lazy("Sigma_inv", lambda: self._dSigma_inv.cpu().numpy())

    @property
    def Lambda_MAP(self):
        if self._post is None:
            return None
        return self._lazy("Lambda_MAP", lambda: self._dense_lambda(self._post.lam_diag.cpu().numpy(),
                                                                   self._post.lam_off.cpu().numpy()))

    @property
    def posterior_covariance(self):
        if self._post is None:
            return None
        if self._post.P is None:     # computed on demand: P = R^T R (one MFMA GEMM)
            self._post = self.eng.posterior(self._dX, self.theta, self.kernel.__name__, self._dSigma_inv,
                                            self.eng.dev(self.fMAP), self.m, want_P=True)
        return self._lazy("P", lambda: self._post.P.cpu().numpy())

    @property
    def posterior_covariance_inv(self):
        if self._post is None:
            return None
        return self._lazy("Pinv", lambda: self.Sigma_inv - self.Lambda_MAP)

    def _dense_lambda(self, diag, off):
        N, mb = self.N, self.m + 1
        L = np.diag(diag)
        idx = np.arange(N)
        pse = idx[idx % mb != 0]
        obs = (pse // mb) * mb
        L[obs, pse] = off[pse]
        L[pse, obs] = off[pse]
        return L

    # ------------------------------------------------------------------ covariance
    def create_Gramian(self, X1, X2, kernel, *args):
        theta = args[0]
        return self.eng.gram(np.asarray(X1), theta, kernel.__name__, self.COVARIANCE_SHRINKAGE).cpu().numpy()

    def create_Gramian_nonsquare(self, X1, X2, kernel, *args):
        return kernel(X1, X2, *args)

    def update_Sigma(self, theta):
        self._dSigma = self.eng.gram(self._dX, theta, self.kernel.__name__, self.COVARIANCE_SHRINKAGE)
        self._invalidate("Sigma")
        self._drop_stale_factor(theta)

    def _drop_stale_factor(self, theta=None):
        """The Cholesky factor kept from update_Sigma_inv belongs to ONE (X, theta): once Sigma has been rebuilt for
        another theta or more rows, prior draws must not use it (an N_old x N_old factor against an
        N-vector, or the factor of the previous theta).  The incremental append keys on _sinv_state itself and is
        left alone: it checks theta and the row prefix before bordering."""
        st = self._sinv_state
        if self._dL is None:
            return
        th = None if theta is None else tuple(float(t) for t in theta)
        if st is None or self._dL.shape[0] != self.N or (th is not None and st[1] != th):
            self._dL_stale = True
        else:
            self._dL_stale = False

    def update_Sigma_inv(self, theta):
        th = tuple(float(t) for t in theta)
        st = self._sinv_state
        done = False
        if (self.incremental and st is not None and self._dLinv is not None and self._dL is not None and st[1] == th
                and st[2] < APPEND_REFRESH and 0 < self.N - st[0].shape[0] <= 64
                and np.array_equal(self.X[:st[0].shape[0]], st[0])):
            try:
                self._dSigma_inv, self._dLinv, self._dL = self.eng.pd_inverse_append(self._dSigma, self._dSigma_inv,
                                                                                       self._dLinv, self._dL)
                self._sinv_state = (self.X.copy(), th, st[2] + 1)
                self.n_appends += 1
                done = True
            except NotPositiveDefinite:
                pass                      # Schur complement lost definiteness to rounding: full inversion below
        if not done:
            # the Cholesky factor of Sigma comes with the inverse: the whitened f_MAP search and the prior draws use it
            if self.incremental:
                self._dSigma_inv, self._dLinv, self._dL = self.eng.pd_inverse_factors3(self._dSigma)
            else:
                (self._dSigma_inv, self._dL), self._dLinv = self.eng.pd_inverse_chol(self._dSigma), None
            self._sinv_state = (self.X.copy(), th, 0)
            self.n_full_inversions += 1
        self._dL_stale = False
        self._invalidate("Sigma_inv", "Pinv")

    def set_theta(self):
        self.theta = self.theta_initial
        if self.theta[1] is None:
            self.theta[1] = 1
        if self.theta[2] is None:
            self.theta[2] = 0.1
        if self.theta[0] is None:
            self.theta[0] = 8

    # ------------------------------------------------------------------ functional T
    def _sinv_dev(self, Sigma_inv_):
        return self._dSigma_inv if Sigma_inv_ is None else self.eng.dev(Sigma_inv_)

    def T(self, f, theta, Sigma_inv_=None):
        T, _ = self.eng.T_and_grad(self._sinv_dev(Sigma_inv_), np.asarray(f, dtype=float).ravel(), self.m, theta[0])
        return T

    def T_grad(self, f, theta, Sigma_inv_=None):
        _, g = self.eng.T_and_grad(self._sinv_dev(Sigma_inv_), np.asarray(f, dtype=float).ravel(), self.m, theta[0])
        return g.cpu().numpy()

    def create_Lambda(self, f, sigma):
        _, _, ld, lo = self.eng.laplace_terms(np.asarray(f, dtype=float).ravel(), self.m, sigma)
        return self._dense_lambda(ld.cpu().numpy(), lo.cpu().numpy())

    def T_hessian(self, f, theta, Sigma_inv_=None):
        Sinv = self.Sigma_inv if Sigma_inv_ is None else np.asarray(Sigma_inv_)
        return -Sinv + self.create_Lambda(f, theta[0])

    def sum_Phi_vec(self, order_of_derivative, f, sigma, over_all_indices=False):
        """gp_model.py:206-218 on the device (ppbo_sum_phi): one value per query, repeated over the query's m+1 rows
        when over_all_indices.  Order 0 is the closed form Phi(Delta/sqrt2) of the Gauss-Hermite integral at :192."""
        if order_of_derivative not in (0, 1, 2):
            print("The derivatives of an order higher than 2 are not needed!")
            return None
        out = self.eng.sum_phi(np.asarray(f, dtype=float).ravel(), self.m, sigma, order_of_derivative).cpu().numpy()
        return np.repeat(out, self.m + 1) if over_all_indices else out

    def sum_Phi(self, i, order_of_derivative, f, sigma, sample_points=None, weights=None):
        """gp_model.py:176-204: the sum for the query whose observation row is i (an element of obs_indices).  The
        quadrature arguments are accepted and unused (closed form)."""
        v = self.sum_Phi_vec(order_of_derivative, f, sigma)
        return None if v is None else float(v[int(i) // (self.m + 1)])

    # ------------------------------------------------------------------ concurrent fits
    def _side_engines(self, n):
        """n extra (Engine, stream) pairs on this model's GPU.  One f_MAP fit is a latency-bound chain of small
        launches that leaves most of the chip idle, so independent fits (evidence at different theta, the random
        restarts of the last iteration) run concurrently: one ppbo_ctx (private workspaces) and one HIP stream per
        host thread; ctypes releases the GIL inside every library call."""
        import torch
        from .engine import Engine
        pool = self.__dict__.setdefault("_side_pool", [])
        while len(pool) < n:
            pool.append((Engine(self.eng.device.index), torch.cuda.Stream(device=self.eng.device)))
        return pool[:n]

    def _run_concurrently(self, jobs, workers):
        """jobs: callables taking an Engine; returns their results in order.  workers <= 1: run on self.eng."""
        import queue
        import threading
        import torch
        if workers <= 1 or len(jobs) <= 1:
            return [job(self.eng) for job in jobs]
        side = self._side_engines(min(workers, len(jobs)))
        main = torch.cuda.current_stream(self.eng.device)
        todo = queue.Queue()
        for k, job in enumerate(jobs):
            todo.put((k, job))
        out, errs = [None] * len(jobs), []

        def worker(eng, stream):
            stream.wait_stream(main)                     # inputs were produced on the caller's stream
            with torch.cuda.stream(stream):
                while True:
                    try:
                        k, job = todo.get_nowait()
                    except queue.Empty:
                        break
                    try:
                        out[k] = job(eng)
                    except Exception as e:           # noqa: BLE001  (re-raised in the caller's thread)
                        errs.append(e)
                stream.synchronize()

        threads = [threading.Thread(target=worker, args=pair) for pair in side]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if errs:
            raise errs[0]
        return out

    @staticmethod
    def _default_workers(N):
        # measured (tools/dev/r4_evidence_workers.py, 20 evidences): N = 512: 76.7 / 32.9 / 26.9 / 27.4 ms with 1 / 4 / 8 / 16
        # contexts; N = 2048: 214.7 / 86.5 / 76.7 / 67.9 ms -- beyond eight the host's launch rate is the limit
        return 8 if N <= 2048 else (4 if N <= 4096 else 2)

    # ------------------------------------------------------------------ evidence / hyper-parameters
    def _evidence_core(self, eng, theta, f0):
        """One Laplace evidence on `eng` from the start vector f0 (device); returns (value incl. log-prior, log-evidence)."""
        import scipy.stats
        Sig = eng.gram(self._dX, theta, self.kernel.__name__, self.COVARIANCE_SHRINKAGE)
        if self.fMAP_method == "whitened":
            Sinv, L = eng.pd_inverse_chol(Sig)
        else:
            Sinv, L = eng.pd_inverse(Sig), None
        fm, st = eng.fit_fmap(Sinv, f0, self.m, theta[0], gtol=1e-4, maxiter=500, L=L)
        _, _, ld, lo = eng.laplace_terms(fm, self.m, theta[0])
        sgn, logdet, _ = eng.laplace_logdet(Sig, ld, lo, self.m)
        log_evidence = st["T"] - 0.5 * sgn * logdet
        lp = (np.log(scipy.stats.lognorm.pdf(theta[0], s=1, scale=np.exp(1)))
              + np.log(scipy.stats.lognorm.pdf(theta[1], s=0.5, scale=np.exp(-1.4)))
              + np.log(scipy.stats.lognorm.pdf(theta[2], s=0.5, scale=np.exp(1.7))))
        return log_evidence + lp, log_evidence

    def evidence(self, theta, f_initial):
        """Laplace log-marginal likelihood + log-prior (gp_model.py:278-319) on the device.
        As in the reference, f_initial is IGNORED and redrawn from N(0, self.Sigma) (:294), the matrix is
        I + Sigma_theta*Lambda_MAP (plus sign, :302) and the determinant term is the sum of sign*logdet over the
        LU factors (:307-310), i.e. sign(prod diag U) * sum log|diag U| with LAPACK's pivoting."""
        theta = [float(t) for t in theta]
        if self.verbose:
            print("---------- Iter results ----------------")
        value, log_evidence = self._evidence_core(self.eng, theta, self._draw_prior())
        if self.verbose:
            print("(scaled) Log-evidence: " + str(log_evidence))
            print("Hyper-parameters: " + str(theta))
        if np.isnan(value) or not np.isfinite(value):
            if self.verbose:
                print("Nan log-evidence!")
            return -500
        if self.verbose:
            print("(scaled) Log-evidence + Log-prior: " + str(value))
        return float(value)

    def evidence_batch(self, thetas, workers=None):
        """evidence() at many thetas: the start vectors are drawn first, in order, from the global stream (so the
        values equal those of sequential evidence() calls), then the independent fits run concurrently on this GPU;
        under torch.distributed (one process per GPU) rank r takes thetas[r::world] and ONE all-gather returns all
        values to every rank.  REQUIREMENT under torch.distributed: every rank calls this with the same thetas and
        an identically seeded global NumPy stream (the start vectors are drawn from it on every rank); a checksum
        of both is compared across ranks first and a mismatch raises.  NaN / inf -> -500 as in the reference
        (gp_model.py:314-316)."""
        from . import dist as _dist
        thetas = [[float(t) for t in th] for th in thetas]
        f0s = [self._draw_prior() for _ in thetas]
        rank, world = _dist.rank_world()
        if world > 1:
            # rank r evaluates thetas[r::world] and the values are merged BY POSITION: that is only meaningful when
            # every rank holds the same candidate list and the same start vectors, i.e. identical NumPy streams
            tarr = np.asarray(thetas, dtype=float)
            _dist.assert_same_across_ranks(
                [len(thetas), float(tarr.sum()), float((tarr * np.arange(1, tarr.size + 1).reshape(tarr.shape)).sum()),
                 float(f0s[0].sum().item()) if f0s else 0.0],
                "evidence_batch: hyper-parameter candidates / start vectors", device=self.eng.device)
        mine = list(range(rank, len(thetas), world))
        workers = self._default_workers(self.N) if workers is None else workers
        jobs = [(lambda eng, k=k: self._evidence_core(eng, thetas[k], f0s[k])[0]) for k in mine]
        vals = self._run_concurrently(jobs, workers)
        vals = [(-500.0 if (v != v or not np.isfinite(v)) else float(v)) for v in vals]
        return _dist.allgather_strided(vals, len(thetas), device=self.eng.device)

    def optimize_theta(self, workers=None):
        """Evidence maximisation over (l, sigma_f) with sigma fixed to 1 (gp_model.py:391-413).  The reference
        drives GPyOpt's Bayesian optimisation (20 initial + 40 iterations = 60 evidence fits); GPyOpt is replaced
        by the same budget of device evidence fits: 20 uniform draws over the reference's box (one concurrent
        batch), then 4 batches of 10 shrinking Gaussian perturbations of the incumbent.  The search TRAJECTORY is
        unpinned (GPyOpt==1.2.6 absent); the objective is the pinned evidence()."""
        if self.verbose:
            print("Hyperparameter optimization begins...")
        start = time.time()
        lo, hi = np.array([0.01, 0.1]), np.array([2.0, 15.0])
        best_v, best_t = -np.inf, None
        self.theta_search_log = []
        for rnd in range(5):
            if rnd == 0:
                cands = lo + np.random.uniform(size=(20, 2)) * (hi - lo)
            else:
                width = 0.25 * (hi - lo) * (0.5 ** (rnd - 1))
                cands = np.clip(best_t + width * np.random.standard_normal((10, 2)), lo, hi)
            vals = self.evidence_batch([[1.0, c[0], c[1]] for c in cands], workers=workers)
            for c, v in zip(cands, vals):
                self.theta_search_log.append((float(c[0]), float(c[1]), float(v)))
                if v > best_v:
                    best_v, best_t = v, c
        if self.verbose:
            print("Optimization of hyperparameters took " + str(time.time() - start) + " seconds.")
        self.theta = [1.0, float(best_t[0]), float(best_t[1])]
        if self.verbose:
            print("The optimized theta is " + str(self.theta))

    # ------------------------------------------------------------------ f_MAP
    def _draw_prior(self):
        """f ~ N(0, Sigma) for the random start (gp_model.py:374,381): L z with the device Cholesky
        factor and z from the global NumPy stream (the reference uses np.random.multivariate_normal)."""
        if self._dL is not None and not self.__dict__.get("_dL_stale", False) and self._dL.shape[0] == self.N:
            # factor of the current Sigma, from update_Sigma_inv
            return self.eng.dgemv(self._dL, np.random.standard_normal(self.N), lower=True)
        key = (self._dSigma.data_ptr(), getattr(self._dSigma, "_version", 0))
        cache = self.__dict__.get("_prior_chol")
        if cache is None or cache[0] != key:
            cache = (key, self.eng.potrf_(self._dSigma.clone()), self._dSigma)     # keeps Sigma alive: the pointer stays unique
            self._prior_chol = cache
        return self.eng.dgemv(cache[1], np.random.standard_normal(self.N), lower=True)

    def update_fMAP(self, random_initial_vector=None, fmap_finding_trials=None, approx_optimization=False):
        trials = self.fMAP_finding_trials if fmap_finding_trials is None else fmap_finding_trials
        rnd = self.fMAP_random_initial_vector if random_initial_vector is None else random_initial_vector
        gtol = 100.0 if approx_optimization else self.fMAP_gtol   # gp_model.py:365-368 (SciPy default gtol 1e-4)
        if self.verbose:
            print("MAP-estimation begins...")
        start = time.time()
        best_T, best = -np.inf, None
        # incremental mode: one warm start replaces the reference's default prior draw (not on the last
        # iteration, whose 10 random restarts are the reference's guard against local optima, gp_model.py:96-97)
        warm = (self.incremental and self.fMAP is not None and len(self.fMAP) <= self.N and not self.last_iteration
                and random_initial_vector is None and fmap_finding_trials is None)
        if warm:
            rnd, trials = False, 1
        starts = []
        for _ in range(trials):
            if self.fMAP is None or rnd or len(self.fMAP) > self.N:
                f0 = self._draw_prior()
            elif len(self.fMAP) < self.N:
                n_old = len(self.fMAP)
                pm = self._post_mean
                if warm and pm is not None and pm.X.shape[0] == n_old:
                    # previous posterior mean at the appended rows: K(X_new, X_old) alpha_old
                    pad = self.eng.predict(pm, self.X[n_old:], score=SCORE_MEAN, want_var=False,
                                           want_best=False)["mu"].cpu().numpy()
                else:                                            # pad with the mean (gp_model.py:375-377)
                    pad = np.full(self.N - n_old, np.mean(self.fMAP))
                f0 = np.concatenate([self.fMAP, pad])
            else:
                f0 = self.fMAP
            starts.append(self.eng.dev(f0))
        # independent restarts (the reference's 10 on the last iteration, gp_model.py:96-97) run concurrently
        t_fit = time.time()
        L = self._dL if self.fMAP_method == "whitened" else None
        jobs = [(lambda eng, f0=f0: eng.fit_fmap(self._dSigma_inv, f0, self.m, self.theta[0], gtol=gtol, L=L))
                for f0 in starts]
        results = self._run_concurrently(jobs, self._default_workers(self.N) if trials > 1 else 1)
        t_fit = (time.time() - t_fit) / max(trials, 1)
        for fm, st in results:
            if trials > 1:               # produced on a side stream, consumed on the caller's from here on
                import torch
                fm.record_stream(torch.cuda.current_stream(self.eng.device))
            self.fit_log.append(dict(N=self.N, method=self.fMAP_method, iterations=st["iterations"], n_cholesky=st["n_cholesky"],
                                     lbfgs_evals=st.get("lbfgs_evals", 0), lbfgs_status=st.get("lbfgs_status", -1),
                                     converged=st["converged"], warm=bool(warm), seconds=t_fit))
            if (self.fMAP_restart_on_stall and not st["converged"] and not approx_optimization
                    and st["gradnorm"] > 1e3 * gtol):
                # opt-in (off by default: it draws from the global NumPy stream, which would shift every later
                # design relative to the reference's trace).  The trust region stalled far from stationarity (no
                # descent predicted / radius collapsed): the reference hands back SciPy's last iterate
                # (gp_model.py:382-389); with the flag set, one fresh start from the prior is tried as well
                print("---!!!--- f_MAP search stopped at |grad T| = " + str(st["gradnorm"]) + "; restarting from a prior draw")
                fm2, st2 = self.eng.fit_fmap(self._dSigma_inv, self._draw_prior(), self.m, self.theta[0], gtol=gtol, L=L)
                if st2["T"] > st["T"] or not np.isfinite(st["T"]):
                    fm, st = fm2, st2
            if self.verbose:
                print("... this took " + str(time.time() - start) + " seconds.")
            if st["T"] > best_T:
                best_T, best, self.fit_stats = st["T"], fm, st
        if best is None:        # every trial ended with a non-finite T: keep the previous estimate, like a failed minimize
            print("---!!!--- f_MAP search produced no finite objective; keeping the previous f_MAP ---!!!---")
            if self.fMAP is None or len(self.fMAP) != self.N:
                self.fMAP = np.zeros(self.N)
            best = self.eng.dev(self.fMAP)
        self.fMAP = best.cpu().numpy()
        # like the reference, the previous Lambda_MAP / posterior covariance stay in place until
        # update_model recomputes them (and survive a failed recomputation, gp_model.py:115-120)
        self._refresh_mean_state(best)

    def _refresh_mean_state(self, fmap_dev):
        from .engine import Posterior
        alpha = self.eng.dgemv(self._dSigma_inv, fmap_dev)
        self._post_mean = Posterior(self.kernel.__name__, tuple(float(t) for t in self.theta), self.m, self._dX, alpha,
                                    None, None, None)

    def _fit_fused(self, defer_posterior=False):
        """The default update (one prior-draw start, whitened search) as ONE library call: ppbo_gp_fit builds Sigma,
        its factor and inverse, runs the search from the prior draw L z0 -- z0 from the global NumPy stream, exactly the
        numbers _draw_prior would have consumed -- and the posterior state, with one host wait instead of one per
        phase (update_Sigma + update_Sigma_inv + update_fMAP + the posterior of src/gp_model.py:91-117).
        Returns False (nothing done) when the configuration needs the call-by-call path."""
        if (self.incremental or self.fMAP_method != "whitened" or self.last_iteration or self.fMAP_finding_trials != 1
                or not self.fMAP_random_initial_vector or self.fMAP_restart_on_stall):
            return False
        if self.verbose:
            print("MAP-estimation begins...")
        start = time.time()
        z0 = np.random.standard_normal(self.N)
        # defer_posterior (update_model, N >= 1024): the call ends with f_MAP; Lambda_MAP / the posterior factor -- a
        # latency-bound chain of 0.9 ms at N = 2048 that mu_star does not need (it reads alpha only) -- then run on a second
        # ctx and stream BESIDE mu_star (_start_posterior / _finish_posterior): update_model at C3 3.9 -> 3.5 ms
        r = self.eng.gp_fit(self._dX, self.theta, self.kernel.__name__, self.m, z0, shrink=self.COVARIANCE_SHRINKAGE,
                            gtol=self.fMAP_gtol, start_is_whitened=True, want_Sigma=True,
                            want_posterior=not defer_posterior)
        th = tuple(float(t) for t in self.theta)
        self._dSigma, self._dSigma_inv, self._dL, self._dLinv = r["Sigma"], r["Sigma_inv"], r["L"], None
        self._sinv_state = (self.X.copy(), th, 0)
        self._dL_stale = False
        self.n_full_inversions += 1
        self._invalidate("Sigma", "Sigma_inv", "Pinv")
        st = r["stats"]
        self.fit_stats = st
        self.fit_log.append(dict(N=self.N, method="whitened (ppbo_gp_fit)", iterations=st["iterations"],
                                 n_cholesky=st["n_cholesky"], lbfgs_evals=st["lbfgs_evals"],
                                 lbfgs_status=st["lbfgs_status"], converged=st["converged"], warm=False,
                                 seconds=time.time() - start))
        fm_host = r["fMAP"].cpu().numpy()
        if not (np.isfinite(st["T"]) and np.all(np.isfinite(fm_host))):
            # neither the search nor the finisher behind it (lbfgs_status 4 = the start had no finite objective; the
            # trust region may still have recovered one: then T and f_MAP ARE finite and the fit stands) produced a finite
            # objective: what update_fMAP does when every trial ends that way -- the previous estimate stays.  Mean AND
            # variance state are rebuilt from it at the CURRENT design, so that next_query never mixes a new mean with a
            # posterior of another N
            print("---!!!--- f_MAP search produced no finite objective; keeping the previous f_MAP ---!!!---")
            if self.fMAP is None or len(self.fMAP) != self.N:
                self.fMAP = np.zeros(self.N)
            kept = self.eng.dev(self.fMAP)
            self._refresh_mean_state(kept)
            try:
                self._post = self._post_mean = self.eng.posterior(self._dX, self.theta, self.kernel.__name__,
                                                                  self._dSigma_inv, kept, self.m, want_P=False)
                self._invalidate("Lambda_MAP", "P", "Pinv")
            except NotPositiveDefinite:
                print("---!!!--- Posterior covariance matrix is not PSD ---!!!---")
                if self._post is not None and self._post.X.shape[0] != self.N:
                    self._post = None      # a posterior of another design: predictions of the variance raise instead
                    self._invalidate("Lambda_MAP", "P", "Pinv")
            return True
        self.fMAP = fm_host
        if self.verbose:
            print("... this took " + str(time.time() - start) + " seconds.")
            print("Current theta is: " + str(self.theta) + " (Acq. = " + str(self.xi_acquisition_function) + ")")
            print("Updating Lambda_MAP and posterior covariance...")
        if defer_posterior:
            self._refresh_mean_state(r["fMAP"])            # alpha: all mu_star needs
            self._start_posterior(r["fMAP"])
        elif r["post"] is not None:
            self._post = self._post_mean = r["post"]
            self._invalidate("Lambda_MAP", "P", "Pinv")
        else:
            print("---!!!--- Posterior covariance matrix is not PSD ---!!!---")   # gp_model.py:119, keep the old one
            self._refresh_mean_state(r["fMAP"])
        return True

    def _start_posterior(self, fmap_dev):
        """Lambda_MAP, alpha and the posterior factor G (ppbo_posterior) on a side ctx / stream, in a host thread of its
        own (the call ends with a host wait; ctypes releases the GIL): the caller goes on with mu_star and collects the
        result with _finish_posterior."""
        import threading
        import torch
        (side, stream), = self._side_engines(1)
        box = {}
        main = torch.cuda.current_stream(self.eng.device)

        # f_MAP and Sigma^-1 were produced on the caller's stream: the dependency is recorded HERE, in the calling thread,
        # before mu_star enqueues anything on `main` -- recorded inside the worker it would land wherever `main` happens to
        # be when that thread first runs, and the side stream would then wait for mu_star's launches as well
        stream.wait_stream(main)

        def work():
            try:
                with torch.cuda.stream(stream):
                    box["post"] = side.posterior(self._dX, self.theta, self.kernel.__name__, self._dSigma_inv, fmap_dev,
                                                 self.m, want_P=False)
                    stream.synchronize()
            except Exception as e:                      # noqa: BLE001  (looked at in _finish_posterior)
                box["err"] = e

        th = threading.Thread(target=work)
        th.start()
        self._pending_posterior = (th, box)

    def _finish_posterior(self):
        pend = self.__dict__.pop("_pending_posterior", None)
        if pend is None:
            return
        import torch
        th, box = pend
        th.join()
        err = box.get("err")
        if isinstance(err, NotPositiveDefinite):
            print("---!!!--- Posterior covariance matrix is not PSD ---!!!---")   # gp_model.py:119, keep the old one
            return
        if err is not None:
            raise err
        post = box["post"]
        main = torch.cuda.current_stream(self.eng.device)
        for t in (post.alpha, post.lam_diag, post.lam_off, post.G):
            t.record_stream(main)                       # allocated on the side stream, consumed on the caller's from here on
        self._post = self._post_mean = post
        self._invalidate("Lambda_MAP", "P", "Pinv")

    # ------------------------------------------------------------------ orchestration (gp_model.py:87-132)
    def update_model(self, optimize_theta=False):
        if self.theta is None:
            self.set_theta()
        init_skip = self.initialization_running and self.skip_computations_during_initialization
        fused = (not init_skip) and (not optimize_theta) and self._fit_fused(defer_posterior=self.N >= 1024)
        if not fused:
            self.update_Sigma(self.theta)
            self.update_Sigma_inv(self.theta)
            if init_skip:
                self.FP.alpha_grid_distribution = "equispaced"
                self.update_fMAP(random_initial_vector=False, fmap_finding_trials=1, approx_optimization=True)
            elif self.last_iteration:
                self.update_fMAP(random_initial_vector=True, fmap_finding_trials=10)
            else:
                self.update_fMAP()
            if optimize_theta:
                self.optimize_theta()
                self.update_fMAP()
                self.update_Sigma(self.theta)
                self.update_Sigma_inv(self.theta)
            if self.verbose:
                print("Current theta is: " + str(self.theta) + " (Acq. = " + str(self.xi_acquisition_function) + ")")
            if not init_skip:
                if self.verbose:
                    print("Updating Lambda_MAP and posterior covariance...")
                start = time.time()
                if self.N >= 1024 and not self.last_iteration:
                    # beside mu_star, as after the fused fit (the restarts of the last iteration occupy the side contexts)
                    self._start_posterior(self.eng.dev(self.fMAP))
                else:
                    try:
                        self._post = self.eng.posterior(self._dX, self.theta, self.kernel.__name__, self._dSigma_inv,
                                                        self.eng.dev(self.fMAP), self.m, want_P=False)
                        self._post_mean = self._post
                        self._invalidate("Lambda_MAP", "P", "Pinv")
                    except NotPositiveDefinite:
                        print("---!!!--- Posterior covariance matrix is not PSD ---!!!---")   # gp_model.py:119, keep the old one
                if self.verbose:
                    print("... this took " + str(time.time() - start) + " seconds.")
        if self.verbose:
            print("Computing mu_star and x_star ...")
        start = time.time()
        try:
            if init_skip and not self.skip_xstaroptimization_during_initialization:
                self.xstar, self.mustar, self.xstars_local = self.mu_star(mustar_finding_trials=1)
            elif self.initialization_running and self.skip_xstaroptimization_during_initialization:
                pass
            elif self.last_iteration:
                self.xstar, self.mustar, self.xstars_local = self.mu_star(mustar_finding_trials=20)
            else:
                self.xstar, self.mustar, self.xstars_local = self.mu_star()
        finally:
            self._finish_posterior()                   # the posterior factor that ran beside mu_star (if any)
        if self.verbose:
            print("... this took " + str(time.time() - start) + " seconds.")

    # ------------------------------------------------------------------ predictions
    def _mean_post(self):
        if self._post_mean is None:
            raise RuntimeError("update_model()/update_fMAP() must run before predictions")
        return self._post_mean

    def mu_Sigma_pred(self, X_pred):
        if self._post is None or self._post.G is None:
            raise RuntimeError("posterior covariance unavailable (skipped during initialisation, gp_model.py:106-107)")
        mu, cov = self.eng.predict_cov(self._post, np.atleast_2d(X_pred), self.COVARIANCE_SHRINKAGE)
        return mu.cpu().numpy(), cov.cpu().numpy()

    def mu_pred(self, X_pred):
        x = np.asarray(X_pred, dtype=float).reshape(1, self.D)
        out = self.eng.predict(self._mean_post(), x, score=SCORE_MEAN, want_var=False, want_best=False)
        return float(out["mu"].item())

    def mu_pred_neq(self, X_pred):
        return -self.mu_pred(X_pred)

    def mu_pred_batch(self, X_pred):
        """Posterior mean of many points in one launch (device tensor in, NumPy out)."""
        out = self.eng.predict(self._mean_post(), X_pred, score=SCORE_MEAN, want_var=False, want_best=False)
        return out["mu"].cpu().numpy()

    # ------------------------------------------------------------------ maximiser of the posterior mean
    def _ascend(self, starts):
        """Projected gradient ascent on the posterior mean from K starts at once (ppbo_mean_ascent: Barzilai-Borwein
        step lengths, monotone safeguard, the whole iteration of a start inside one workgroup).  SURVEY 8(f) f-2."""
        xs, mus, _ = self.eng.mean_ascent(self._mean_post(), np.clip(np.atleast_2d(starts).astype(float), 0.0, 1.0),
                                          iters=ASCENT_ITERS, tol=1e-9)
        return xs.cpu().numpy(), mus.cpu().numpy()

    def _polish(self, x0):
        """Bounded quasi-Newton polish of one point with the analytic device gradient."""
        post = self._mean_post()

        def fg(x):
            mu, g = self.eng.mean_grad(post, np.clip(x, 0.0, 1.0)[None, :])
            return -float(mu.cpu().numpy()[0]), -g.cpu().numpy()[0]

        res = scipy.optimize.minimize(fg, x0, jac=True, method="L-BFGS-B", bounds=self.bounds,
                                      options={"maxiter": 200, "ftol": 1e-15, "gtol": 1e-10})
        return np.clip(res.x, 0.0, 1.0), -float(res.fun)

    def _candidate_pool(self):
        """SEARCH_CANDIDATES uniform points, drawn ONCE from the global NumPy stream and kept resident in HBM; every
        mu_star trial sees them through a fresh random rotation frac(pool + shift) (ppbo_shift_points), which
        keeps them uniform -- instead of generating and uploading 65536 x D numbers per trial."""
        pool = self.__dict__.get("_pool")
        if pool is None or pool.shape[1] != self.D:
            pool = self.eng.dev(np.random.uniform(0.0, 1.0, (SEARCH_CANDIDATES, self.D)))
            self._pool = pool
        return pool

    def mu_star(self, mustar_finding_trials=None):
        """argmax of the posterior mean (gp_model.py:415-437).  The reference runs SciPy differential evolution
        `trials` times (2 k ... 17 k sequential mu_pred calls each); here a trial is ONE device enqueue
        (ppbo_mean_search): score 65536 rotated pool candidates plus the design points and the previous x*, keep the
        ASCENT_STARTS best that are > 0.05 apart, run the whole Barzilai-Borwein ascent of each inside one kernel;
        the host reads ASCENT_STARTS x (D + 1) numbers back and polishes the winner.  All distinct converged maxima
        feed xstars_local (gp_model.py:430-431)."""
        import torch
        trials = self.mustar_finding_trials if mustar_finding_trials is None else mustar_finding_trials
        D, N = self.D, self.N
        post = self._mean_post()
        pool = self._candidate_pool()
        # ALL trials in one enqueue (ppbo_mean_search_multi): each sees the resident pool through its own rotation, formed
        # on the fly; one screening launch, one thinning launch, one start-selection launch (a workgroup per trial), one
        # ascent launch of trials x ASCENT_STARTS workgroups.  (Rounds 3-4 ran one ppbo_mean_search per trial on three
        # stream / ctx lanes: 3 trials at C3 2.0-2.3 ms, their kernels mostly serialised.)
        # The design points (where f_MAP's maxima sit) and the previous x* join the first trial only: they would claim
        # the same ASCENT_STARTS basins in every trial and leave the uniform candidates' basins unexplored.
        shifts = np.stack([np.random.uniform(0.0, 1.0, D) for _ in range(trials)]) if trials > 0 else np.zeros((0, D))
        xprev = np.asarray(self.xstar if self.xstar is not None else self.X[0], dtype=float).reshape(D)
        queued = []
        for t0 in range(0, trials, 64):                                # the entry takes up to 64 trials
            xs_d, mu_d = self.eng.mean_search_multi(post, pool, shifts[t0:t0 + 64], "design" if t0 == 0 else None,
                                                    xprev if t0 == 0 else None, K=ASCENT_STARTS, sep=5e-2,
                                                    iters=ASCENT_ITERS, tol=1e-9, screen_fp32=self.mustar_screen_fp32)
            queued.append((xs_d, mu_d))
        found, winners, winner_mu = [], [], {}
        if queued:
            all_x = torch.cat([q[0] for q in queued]).cpu().numpy()              # [trials, K, D]
            all_v = torch.cat([q[1] for q in queued]).cpu().numpy()              # [trials, K], -inf = no start
            per_trial = []
            for t in range(trials):
                ok = np.isfinite(all_v[t])
                xs, vals = all_x[t][ok], all_v[t][ok]
                per_trial.append((xs, vals))
                if len(vals):
                    winners.append((t, int(np.argmax(vals))))
            if winners:
                # the ascent stops on |projected gradient| * step < 1e-9; the quasi-Newton polish (a device round
                # trip per function value) is only worth its milliseconds when a winner is NOT yet stationary
                wx = np.stack([per_trial[t][0][b] for t, b in winners])
                mw, gw = self.eng.mean_grad(post, wx)
                mg = torch.cat([mw.reshape(-1, 1), gw], dim=1).cpu().numpy()       # one read-back: mu | grad per winner
                def stationary(x, gb, v):
                    pg = np.where(((x <= 0.0) & (gb < 0.0)) | ((x >= 1.0) & (gb > 0.0)), 0.0, gb)
                    return np.abs(pg).max() <= POLISH_GRAD_TOL * max(abs(v), 1e-300)
                todo = []
                for k, ((t, b), row) in enumerate(zip(winners, mg)):
                    xs, vals = per_trial[t]
                    if stationary(xs[b], row[1:], vals[b]):
                        # mu_pred at this point, already on the host (the same K*' alpha in fp64 by mean_grad_kernel): if it
                        # ends up as x*, mustar needs no further device round trip
                        winner_mu[xs[b].tobytes()] = float(row[0])
                    else:
                        todo.append((t, b))
                if todo:
                    # winners the 100-evaluation ascent left short of stationarity: ONE more device ascent for all of them
                    # (four times the budget, a tighter step tolerance) before anything goes through SciPy -- the
                    # quasi-Newton polish costs a device round trip per function value
                    self.polish_log["device_reascents"] += len(todo)
                    p0 = np.stack([per_trial[t][0][b] for t, b in todo])
                    xa, ma, _ = self.eng.mean_ascent(post, p0, iters=4 * ASCENT_ITERS, tol=1e-12)
                    ma2, ga = self.eng.mean_grad(post, xa)
                    xa, rows = xa.cpu().numpy(), torch.cat([ma2.reshape(-1, 1), ga], dim=1).cpu().numpy()
                    for (t, b), xk, row in zip(todo, xa, rows):
                        xs, vals = per_trial[t]
                        if row[0] >= vals[b]:
                            xs[b], vals[b] = xk, float(row[0])
                        if stationary(xs[b], row[1:], vals[b]) and row[0] >= vals[b]:
                            winner_mu[xs[b].tobytes()] = float(row[0])
                            continue
                        self.polish_log["scipy_polishes"] += 1
                        xp, vp = self._polish(xs[b])
                        if vp >= vals[b]:
                            xs[b], vals[b] = xp, vp
            for xs, vals in per_trial:
                found.extend(zip(vals.tolist(), xs))
        if not found:                 # no finite mean anywhere (cannot happen with a fitted model): keep the old x*
            x0 = self.xstar if self.xstar is not None else self.X[0]
            return np.asarray(x0, dtype=float).reshape(D,), self.mu_pred(x0), np.asarray(x0, dtype=float).reshape(1, D)
        found.sort(key=lambda p: -p[0])
        xstar = found[0][1].copy()
        # distinct maxima > 0.1 apart, best first (gp_model.py:430-431): one distance matrix, then a greedy sweep that
        # strikes everything within 0.1 of a kept maximum (the pairwise Python loop this replaces cost 10 ms at D = 20,
        # where 3 trials return ~90 converged points and ~45 distinct maxima: more than the searches themselves)
        P = np.stack([x for _, x in found])
        sq = (P * P).sum(axis=1)
        d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * (P @ P.T), 0.0)
        struck = np.zeros(len(P), dtype=bool)
        keep = []
        for i in range(len(P)):
            if not struck[i]:
                keep.append(i)
                struck |= d2[i] <= 1e-2
        xstars_local = P[keep].reshape(-1, D)
        mustar = winner_mu.get(xstar.tobytes())
        return xstar.reshape(D,), (self.mu_pred(xstar) if mustar is None else mustar), xstars_local
