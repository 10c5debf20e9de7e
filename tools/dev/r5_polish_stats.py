"""How often mu_star's winners need more than the 100-evaluation ascent: device re-ascents and SciPy polishes over the
model fixtures (3 trials x 5 calls each) and over a 25-query C1 loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
for cfg in ("smoke", "rq", "cam_small", "c2", "c4", "c3"):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
    D, m = int(g["D"]), int(g["m"])
    st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD",
                       theta_initial=list(map(float, g["theta"])), m=m, verbose=False, kernel=str(g["kernel"]))
    gp = GPModel(st)
    np.random.seed(0)
    gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
    gp.update_model()
    for _ in range(5):
        gp.mu_star()
    print(cfg, "N", gp.N, "D", D, gp.polish_log, "mustar", gp.mustar, flush=True)
