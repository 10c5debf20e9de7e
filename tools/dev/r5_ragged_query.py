"""The other pieces of a query at the reference's default star size (m = 25: N = 26 n_q) against the aligned m = 31
shape of the BASELINE configs: cold fit, line acquisitions, mu_star, next_query.
python tools/dev/r5_ragged_query.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from r5_ragged_time import design
from ppbo_amd.engine import get_engine
from ppbo_amd.gp_model import GPModel
from ppbo_amd.acquisition import next_query

def med(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

eng = get_engine(0)
D = 20
for (n_q, m) in ((64, 31), (80, 25), (79, 25)):
    X, st = design(D, n_q, m)
    N = X.shape[0]
    th = [0.09, 0.3, 0.5]
    z0 = np.random.default_rng(3).standard_normal(N)
    fit = lambda: eng.gp_fit(X, th, "SE_kernel", m, z0, start_is_whitened=True)
    r = fit()
    t_fit = med(fit)
    post = r["post"]
    B, G, S = 512, 70, 150
    rng = np.random.default_rng(5)
    xi = np.zeros((B, D)); xi[np.arange(B), rng.integers(0, D, B)] = 1.0
    x = rng.random((B, D)) * (xi == 0)
    al = np.linspace(0.005, 0.995, G)
    z = rng.standard_normal((S, G))
    t_line = med(lambda: eng.line_acq_xi(post, xi, x, al, z, 0.1))
    st.xi_acquisition_function = "EI-EXT"
    gp = GPModel(st)
    np.random.seed(0)
    rows = []
    for q in range(n_q):
        e = np.zeros(D); e[q % D] = 1.0
        xx = np.random.rand(D); xx[q % D] = 0.0
        a = np.random.rand()
        rows.append(np.concatenate([a * e + xx, e, [a]]))
    gp.update_feedback_processing_object(np.array(rows)); gp.update_data(); gp.turn_initialization_off()
    t_upd = med(lambda: gp.update_model(), 3)
    t_nq = med(lambda: next_query(st, gp), 3)
    print(f"N={N} (n_q={n_q}, m={m}): gp_fit {t_fit:.2f} ms ({r['stats']['lbfgs_evals']} evals)  line_acq 512x70x150 {t_line:.2f} ms  "
          f"update_model {t_upd:.2f} ms  next_query EI-EXT {t_nq:.2f} ms", flush=True)
