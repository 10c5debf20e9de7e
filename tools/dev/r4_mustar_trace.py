"""mu_star at the C3 shape: wall per trial and (under rocprofv3 --kernel-trace) the kernels of a trial."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
D, m = int(g["D"]), int(g["m"])
st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD",
                   theta_initial=list(map(float, g["theta"])), m=m, verbose=False, kernel=str(g["kernel"]))
gp = GPModel(st)
np.random.seed(0)
gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
gp.set_theta(); gp._fit_fused()
gp.mu_star(mustar_finding_trials=1)
for trials in (1, 3, 8):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, v, loc = gp.mu_star(mustar_finding_trials=trials)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{cfg} mu_star trials={trials}: {np.median(ts):.2f} ms total, {np.median(ts) / trials:.2f} ms per trial; mustar {v:.6f}, {len(loc)} local maxima")
