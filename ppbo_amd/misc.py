"""src/misc.py for the drop-in.  Host-side helpers (O(D) work, no GPU): interval of a projective line inside the
box (src/misc.py:27-61), the box corners used for initial queries (src/misc.py:143-147), the two normal densities.
The matrix helpers the GP path calls -- regularize_covariance (src/misc.py:71-88), pd_inverse (:96-100),
is_positive_definite (:120-126) -- keep the reference's names and NumPy-in / NumPy-out signatures and run on the
device through the C-ABI (there is no host implementation behind them)."""
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


def regularize_covariance(X, reg_level=1e-4, pos_diag=True, jitter=1e-7):
    """src/misc.py:71-88 on the device (ppbo_regularize_covariance): negative diagonal -> jitter when pos_diag,
    shrink toward tr(X)/n I; the SVD round trip is the identity and is not executed."""
    from .engine import get_engine
    return get_engine().regularize_covariance(np.asarray(X, dtype=float), reg_level, pos_diag, jitter).cpu().numpy()


def pd_inverse(matrix):
    """Inverse of a positive definite matrix (src/misc.py:96-100) by the device Cholesky + triangular inverse;
    raises ppbo_amd.engine.NotPositiveDefinite where SciPy raises LinAlgError."""
    from .engine import get_engine
    return get_engine().pd_inverse(np.asarray(matrix, dtype=float)).cpu().numpy()


def is_positive_definite(M):
    """src/misc.py:120-126: does the (device) Cholesky factorization succeed?"""
    from .engine import NotPositiveDefinite, get_engine
    eng = get_engine()
    try:
        eng.potrf_(eng.dev(np.asarray(M, dtype=float)).clone())
        return True
    except NotPositiveDefinite:
        print('Function is_positive_definite: Matrix is not positive definite!')
        return False
