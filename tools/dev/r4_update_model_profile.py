"""Host-side profile of GPModel.update_model at the C3 shape (the drop-in's fit: ppbo_gp_fit + the Python around it)."""
import os, sys, time, cProfile, pstats, io
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
gp.set_theta()
for _ in range(3): gp.update_model()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); gp.update_model(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{cfg} update_model: median {np.median(ts):.3f} ms, min {min(ts):.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): gp.update_model()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
