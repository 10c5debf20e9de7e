"""Probe (not a test): the random-model sweep of tests/test_gpu_whitened.py::test_random_models_..., printing one line
per case instead of asserting -- T of both paths, their distance, gradient norms, evaluation counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_whitened import _random_case, host  # noqa: E402


def main(n):
    from ppbo_amd.engine import get_engine
    eng = get_engine(0)
    for seed in range(n):
        X, m, kernel, theta = _random_case(seed)
        N, sig = X.shape[0], theta[0]
        S = eng.gram(X, theta, kernel)
        Sinv, L = eng.pd_inverse_chol(S)
        f_init = host(eng.dgemv(L, np.random.default_rng(seed).standard_normal(N), lower=True))
        fw, sw = eng.fit_fmap(Sinv, f_init, m, sig, gtol=1e-6, L=L)
        ft, stt = eng.fit_fmap(Sinv, f_init, m, sig, gtol=1e-6)
        d = np.abs(host(fw) - host(ft)).max() / max(np.abs(host(ft)).max(), 1e-300)
        flag = "DIFF" if d > 1e-4 else "    "
        print(f"{seed:3d} {flag} D={X.shape[1]:2d} N={N:4d} m={m:2d} {kernel[:2]} sig={theta[0]:.4f} l={theta[1]:.3f} sf={theta[2]:.3f} | "
              f"Tw={sw['T']:.6f} Tt={stt['T']:.6f} rel|df|={d:.1e} gw={sw['gradnorm']:.1e} gt={stt['gradnorm']:.1e} "
              f"evals={sw['lbfgs_evals']} st={sw['lbfgs_status']} chol_w={sw['n_cholesky']} chol_t={stt['n_cholesky']} it_t={stt['iterations']}",
              flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 48)
