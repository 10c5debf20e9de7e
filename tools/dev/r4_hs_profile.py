"""cProfile of one Hsampler cycle and of mu_star at the C3 shape (host-side hot spots)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
from ppbo_amd.random_fourier_sampler import Hsampler
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
D, m = int(g["D"]), int(g["m"])
st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD",
                   theta_initial=list(map(float, g["theta"])), m=m, verbose=False, kernel=str(g["kernel"]))
gp = GPModel(st)
np.random.seed(0)
gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
gp.set_theta(); gp._fit_fused()
gp.xstar, gp.mustar, gp.xstars_local = gp.mu_star()
F = 4096 if cfg == "c3" else 1000
def cycle():
    hs = Hsampler(gp, F)
    hs.generate_basis(); hs.update_phi_X(); hs.update_omega_MAP(); hs.update_covariancematrix(); hs.sample_xstar()
    return hs
cycle()
for name, fn in (("hsampler cycle", cycle), ("mu_star(3)", lambda: gp.mu_star())):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{cfg} {name}: {np.median(ts):.2f} ms")
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
