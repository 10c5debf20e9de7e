"""Covariance functions with the reference's call signature (src/kernels.py:19-53), evaluated
by the HIP cross-covariance kernel (ppbo_cross_cov).  Inputs/outputs are NumPy arrays; the
function objects carry the reference's __name__ because Hsampler dispatches on it
(src/random_fourier_sampler.py:27,40)."""
from __future__ import annotations

import numpy as np

from .engine import get_engine


def _eval(name, X1, X2, theta):
    if theta[1] <= 0 or theta[2] <= 0:
        print("Check hyperparameter values!")          # the reference only prints (src/kernels.py:22-23)
    X1 = np.atleast_2d(np.asarray(X1, dtype=np.float64))
    X2 = np.atleast_2d(np.asarray(X2, dtype=np.float64))
    return get_engine().cross_cov(X1, X2, theta, name).cpu().numpy()


def SE_kernel(X1, X2, theta):
    return _eval("SE_kernel", X1, X2, theta)


def RQ_kernel(X1, X2, theta):
    return _eval("RQ_kernel", X1, X2, theta)


def camphor_copper_kernel(X1, X2, theta):
    return _eval("camphor_copper_kernel", X1, X2, theta)


BY_NAME = {f.__name__: f for f in (SE_kernel, RQ_kernel, camphor_copper_kernel)}
