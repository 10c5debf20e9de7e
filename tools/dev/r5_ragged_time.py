"""Scoring rate (mean + variance + EI + argmax) at shapes OFF the aligned fast path: the reference's default m = 25
(N = 26 n_q, star blocks of 26 rows), candidate counts that are not multiples of 128, N not a multiple of 128.
python tools/dev/r5_ragged_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_POINTWISE_EI
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings

def design(D, n_q, m, seed=0):
    np.random.seed(seed)
    rows = []
    for q in range(n_q):
        xi = np.zeros(D); xi[q % D] = 1.0
        x = np.random.rand(D); x[q % D] = 0.0
        a = np.random.rand()
        rows.append(np.concatenate([a * xi + x, xi, [a]]))
    st = PPBO_settings(D=D, bounds=((0.0, 1.0),) * D, xi_acquisition_function="PCD", theta_initial=[0.09, 0.3, 0.5], m=m,
                       verbose=False)
    gp = GPModel(st)
    gp.update_feedback_processing_object(np.array(rows)); gp.update_data()
    return np.asarray(gp.X), st

def main():
  eng = get_engine(0)
  D = 20
  for (n_q, m, M) in ((64, 31, 65536), (64, 31, 65000), (79, 25, 65536), (80, 25, 65536), (80, 25, 65000), (63, 31, 65536), (40, 25, 16384)):
      X, st = design(D, n_q, m)
      N = X.shape[0]
      th = [0.09, 0.3, 0.5]
      z0 = np.random.default_rng(3).standard_normal(N)
      r = eng.gp_fit(X, th, "SE_kernel", m, z0, start_is_whitened=True)
      post = r["post"]
      Xc = eng.dev(np.random.default_rng(1).random((M, D)))
      for _ in range(3):
          o = eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1, want_mu=False, want_var=False)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      reps = 10
      for _ in range(reps):
          o = eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1, want_mu=False, want_var=False)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
      flops = M * N * (N / 2.0) * 2
      print(f"N={N} (n_q={n_q}, m={m}) M={M}: {dt * 1e3:.3f} ms/step  {M / dt:.3e} evals/s  triangular-equivalent {flops / dt / 1e12:.1f} TF  fit evals {r['stats']['lbfgs_evals']}", flush=True)

if __name__ == "__main__":
    main()
