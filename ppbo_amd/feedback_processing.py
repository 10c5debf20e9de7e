"""Design-matrix shaping on the host (the hot path's input contract).

Mirrors FeedbackProcessing of the reference (src/feedback_processing.py): every query row
[alpha*xi + x ; xi ; alpha*] becomes one observation row followed by m pseudo-observation rows
on the same projective line; X is scaled to [0,1]^D.  This is O(N D) NumPy work and stays on
the host (out of scope as a GPU target); its output layout is what the kernels
consume: N = n_q (m+1), row q(m+1) = observation of query q.
"""
from __future__ import annotations

import numpy as np
import scipy.stats

from .misc import alpha_bounds


def _tgn_sample(size, gamma, loc, a, b, rng=np.random):
    """Truncated generalised-normal draws on [a, b] (src/TGN_distribution.py:21-25 defines the
    density: gennorm with shape gamma and scale Gamma(gamma)|b-a|/10).  The reference samples it
    by adaptive rejection (arspy); here by inverting the truncated CDF, which is exact."""
    from scipy.special import gamma as Gamma
    scale = Gamma(gamma) * abs(b - a) / 10.0
    dist = scipy.stats.gennorm(gamma, loc=loc, scale=scale)
    lo, hi = dist.cdf(a), dist.cdf(b)
    u = lo + (hi - lo) * rng.uniform(size=size)
    return dist.ppf(u)


class FeedbackProcessing:
    def __init__(self, D, m, original_bounds, alpha_grid_distribution, TGN_speed):
        self.D, self.m = D, m
        self.original_bounds = original_bounds
        self.bounds = ((0, 1),) * D
        self.alpha_grid_distribution = alpha_grid_distribution
        self.TGN_speed = TGN_speed
        self.iter_number = 1
        self.X_obs = self.X_full = self.X = self.N = None
        self.obs_indices = self.pseudobs_indices = self.latest_obs_indices = None
        self._lo = np.array([b[0] for b in original_bounds], dtype=float)
        self._hi = np.array([b[1] for b in original_bounds], dtype=float)

    # ---- data flow -------------------------------------------------------------
    def initialize_data(self, X_obs):
        self.X_obs = np.asarray(X_obs, dtype=float)
        self.create_X()
        self.create_indices_bookkeeping()

    def update_data(self, X_obs):
        self.iter_number += 1
        self.X_obs = np.asarray(X_obs, dtype=float)
        self.update_X()
        self.create_indices_bookkeeping()

    def create_X(self):
        """X_full / X / N from every row of X_obs (src/feedback_processing.py:110-130)."""
        self.X_full = np.vstack([self._query_block(r) for r in self.X_obs])
        self._finish()

    def update_X(self):
        """X_obs has ONE more row than X_full covers: append its block (src/feedback_processing.py:133-154)."""
        self.X_full = np.vstack([self.X_full, self._query_block(self.X_obs[-1])])
        self._finish()

    def _finish(self):
        self.X = self.scale(self.X_full[:, :self.D])
        self.N = self.X.shape[0]

    def _query_block(self, row):
        D, m = self.D, self.m
        point, xi, alpha_star = row[:D], row[D:2 * D], row[-1]
        x = np.where(xi == 0, point, 0.0)
        grid = self.xi_grid(xi=xi, x=x, alpha_star=alpha_star)
        blk = np.empty((m + 1, 2 * D + 1))
        blk[0, :D] = point
        blk[1:, :D] = grid
        blk[:, D:2 * D] = xi
        blk[0, -1], blk[1:, -1] = 0.0, 1.0       # is-pseudo-observation flag
        return blk

    # ---- pseudo-observation grids ----------------------------------------------
    def xi_grid(self, xi, x=None, alpha_grid_distribution=None, alpha_star=None, m=None, is_scaled=False):
        dist = self.alpha_grid_distribution if alpha_grid_distribution is None else alpha_grid_distribution
        m = self.m if m is None else m
        if is_scaled:
            a_lo, a_hi = 0.0, 1.0
        else:
            a_lo, a_hi = alpha_bounds(xi, self._lo, self._hi)
        span = abs(a_hi - a_lo)
        alpha = np.empty(0)
        while alpha.size != m:                    # redraw until m distinct values (reference does the same)
            if dist == "equispaced":
                eps = (a_hi - a_lo) * 0.005       # half of noise_level 0.01 keeps points off the boundary
                alpha = np.linspace(a_lo + eps, a_hi - eps, num=m) + np.random.normal(0, span * 0.01, m)
            elif dist == "Cauchy":
                alpha = scipy.stats.cauchy.rvs(loc=float(alpha_star), scale=span * 0.07, size=m)
            elif dist == "TGN":
                gamma = 3.0 / np.power(max(self.iter_number + 1 - self.D, 1), self.TGN_speed) + 2.0
                alpha = _tgn_sample(m, gamma, float(alpha_star), a_lo, a_hi)
            else:
                print("Uknown alpha-distribution: " + str(dist))
                raise ValueError(dist)
            alpha = np.unique(np.clip(alpha, a_lo, a_hi))
        xi = np.asarray(xi, dtype=float).reshape(1, self.D)
        grid = alpha.reshape(m, 1) * xi
        if x is None:
            return grid[:, ~(grid == 0).all(axis=0)]
        return grid + np.asarray(x, dtype=float)[None, :]

    # ---- bookkeeping -------------------------------------------------------------
    def is_pseudobs(self, i):
        return bool(self.X_full[i, 2 * self.D])

    def create_indices_bookkeeping(self):
        flag = self.X_full[:, 2 * self.D].astype(bool)
        idx = np.arange(self.N)
        self.obs_indices = idx[~flag].tolist()
        self.pseudobs_indices = idx[flag].tolist()
        last = np.maximum.accumulate(np.where(~flag, idx, -1))
        self.latest_obs_indices = last.tolist()

    # ---- scaling -----------------------------------------------------------------
    def scale(self, X, retain_0_values=False):
        X = np.asarray(X, dtype=float)
        out = (X - self._lo) / np.abs(self._hi - self._lo)
        if retain_0_values:
            out[X == 0] = 0
        return out

    def unscale(self, X, retain_0_values=False):
        X = np.asarray(X, dtype=float)
        out = X * np.abs(self._hi - self._lo) + self._lo
        if retain_0_values:
            out[X == 0] = 0
        return out
