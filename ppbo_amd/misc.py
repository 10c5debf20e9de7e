"""Host-side helpers of the drop-in (O(D) work, no GPU): interval of a projective line inside the
box (src/misc.py:27-61) and the box corners used for initial queries (src/misc.py:143-147)."""
from __future__ import annotations

import itertools

import numpy as np


def alpha_bounds(xi, lower, upper):
    """[alpha_min, alpha_max] such that alpha*xi stays inside [lower, upper] coordinate-wise."""
    xi, lower, upper = (np.asarray(a, dtype=float) for a in (xi, lower, upper))
    pos, neg = xi > 0, xi < 0
    lo_c = np.concatenate([lower[pos] / xi[pos], upper[neg] / xi[neg]])
    hi_c = np.concatenate([lower[neg] / xi[neg], upper[pos] / xi[pos]])
    a_lo = lo_c.max() if lo_c.size else -np.inf
    a_hi = hi_c.min() if hi_c.size else np.inf
    if a_lo > a_hi:
        print("Error: alpha_min > alpha_max!")
    if a_lo == -np.inf:
        print("Error: alpha_min is -infinity!")
    if a_hi == np.inf:
        print("Error: alpha_max is infinity!")
    return a_lo, a_hi


def hypercube_corners(bounds):
    return np.array(list(itertools.product(*[(b[0], b[1]) for b in bounds])))


def var2_normal_pdf(x):
    return np.exp(-0.25 * np.square(x)) / np.sqrt(4.0 * np.pi)


def std_normal_pdf(x):
    return np.exp(-0.5 * np.square(x)) / np.sqrt(2.0 * np.pi)
