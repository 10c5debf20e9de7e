"""next_query at the C3 shape: wall time against the device time of its line-acquisition kernels (library event brackets)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.acquisition import next_query
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
D, m, th, kern = int(g["D"]), int(g["m"]), g["theta"], str(g["kernel"])
for acq in ("EI-EXT", "EI", "EI-VARMAX"):
    st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function=acq,
                       theta_initial=list(map(float, th)), m=m, verbose=False, kernel=kern)
    gp = GPModel(st)
    np.random.seed(0)
    gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
    gp.update_model()
    next_query(st, gp); next_query(st, gp)
    eng = gp.eng
    eng.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    next_query(st, gp)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    tot = {k: eng.profile_read(k) for k in ("line_kstar", "line_y", "line_cov", "line_mc", "kstar", "quadform", "score")}
    eng.profile(False)
    dev = sum(v[0] for v in tot.values())
    print(f"{cfg} {acq}: wall {wall:.2f} ms (with brackets), bracketed kernels {dev:.2f} ms: " +
          " ".join(f"{k} {v[0]:.2f}/{v[1]}" for k, v in tot.items() if v[1]))
    pr = cProfile.Profile(); pr.enable(); next_query(st, gp); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:1800])
