"""One whole PPBO query at the C3 shape through the drop-in objects -- update_model (fused fit + mu_star), next_query,
one Hsampler cycle -- warmed, then once more with wall-clock marks; run under `rocprofv3 --kernel-trace` the LAST
occurrence of each phase is the traced one (tools/dev/trace_summary.py).   python tools/dev/r5_query_trace.py [cfg] [acq]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.acquisition import next_query
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
from ppbo_amd.random_fourier_sampler import Hsampler
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
acq = sys.argv[2] if len(sys.argv) > 2 else "EI-EXT"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
D, m, th, kern = int(g["D"]), int(g["m"]), g["theta"], str(g["kernel"])
st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function=acq,
                   theta_initial=list(map(float, th)), m=m, verbose=False, kernel=kern)
gp = GPModel(st)
np.random.seed(0)
gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
F = 4096 if cfg in ("c3", "c5") else 1000
def hs_cycle():
    hs = Hsampler(gp, F)
    hs.generate_basis(); hs.update_phi_X(); hs.update_omega_MAP(); hs.update_covariancematrix(); hs.sample_xstar()
phases = (("update_model", lambda: gp.update_model()), ("next_query " + acq, lambda: next_query(st, gp)),
          ("hsampler_cycle", hs_cycle))
for rep in range(3):
    for name, fn in phases:
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        if rep == 2:
            print(f"{cfg} {name}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
        time.sleep(0.002)   # a visible gap between the phases in the timeline
