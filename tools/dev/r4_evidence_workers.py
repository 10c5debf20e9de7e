"""evidence_batch of 20 hyper-parameter candidates against the number of concurrent contexts."""
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
gp.set_theta(); gp.update_model()
rng = np.random.default_rng(5)
th0 = np.asarray(g["theta"], dtype=float)
thetas = [[th0[0], th0[1] * rng.uniform(0.5, 2.0), th0[2] * rng.uniform(0.5, 2.0)] for _ in range(20)]
for w in (1, 2, 4, 8, 16, 20):
    np.random.seed(1); gp.evidence_batch(thetas[:w], workers=w)          # contexts created, workspaces grown
    np.random.seed(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = gp.evidence_batch(thetas, workers=w)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{cfg} workers {w:2d}: 20 evidences in {dt * 1e3:7.1f} ms = {dt * 1e3 / 20:.2f} ms each; checksum {float(np.sum(v)):.9e}")
