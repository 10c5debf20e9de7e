"""512 lines x 70 points x 150 draws at C3 a few times (target for rocprofv3 --kernel-trace) + next_query wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "c3.npz")))
X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
r = eng.gp_fit(X, th, kern, m, g["f_init"])
post = r["post"]
D = X.shape[1]
B, G, S = 512, 70, 150
rng = np.random.default_rng(6)
xis = np.eye(D)[np.arange(B) % D]
xs = rng.random((B, D)); xs[np.arange(B), np.arange(B) % D] = 0.0
al = np.linspace(0.005, 0.995, G)
z = eng.dev(rng.standard_normal((S, G)))
xd, sd, ad = eng.dev(xis), eng.dev(xs), eng.dev(al)
mustar = float(np.max(g["mu"]))
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ei, vm = eng.line_acq_xi(post, xd, sd, ad, z, mustar, jitter=1e-10 * float(th[2]) ** 2)
    torch.cuda.synchronize()
    print(f"rep {rep}: line_acq_xi {B} x {G} x {S}: {(time.perf_counter() - t0) * 1e3:.3f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "nq":
    from ppbo_amd.acquisition import next_query
    from ppbo_amd.gp_model import GPModel
    from ppbo_amd.ppbo_settings import PPBO_settings
    for acq in ("EI-EXT-FAST", "EI-EXT", "EI", "EXR", "EI-VARMAX"):
        st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function=acq,
                           theta_initial=list(map(float, th)), m=m, verbose=False, kernel=kern)
        gp = GPModel(st)
        np.random.seed(0)
        gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
        gp.update_model()
        next_query(st, gp)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); next_query(st, gp); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"next_query {acq} at C3: {np.median(ts):.2f} ms (median of 5)")
