"""C2 / C1-sized fits: the one-launch whitened search (PPBO_LBFGS_LOOP=1) against the slot form (=0): time, evaluations, f_MAP."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import Engine
res = {}
for mode in (0, 1):
    os.environ["PPBO_LBFGS_LOOP"] = str(mode)
    eng = Engine(0)
    for name in ("smoke", "c2", "rq"):
        g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{name}.npz")))
        X = eng.dev(g["X"]); m = int(g["m"]); th = g["theta"]; kern = str(g["kernel"])
        z0 = eng.dev(np.random.default_rng(2).standard_normal(X.shape[0]))
        ts = []
        for rep in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        f = r["fMAP"].cpu().numpy()
        res[mode, name] = f
        print(f"loop={mode} {name} N={X.shape[0]}: fit {np.median(ts[2:]) * 1e3:.3f} ms  evals {r['stats']['lbfgs_evals']} status {r['stats']['lbfgs_status']} T {r['stats']['T']:.12f}", flush=True)
    eng.close()
for name in ("smoke", "c2", "rq"):
    a, b = res[0, name], res[1, name]
    print(name, "max |f_loop - f_slots| / max|f| =", np.abs(a - b).max() / np.abs(a).max())
