"""Phase stamps of the whitened search's judgement kernel (lbfgs_step_kernel, verbose = 2) at C3 and C2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ppbo_amd.engine import get_engine
eng = get_engine(0)
for name in ("c3", "c2"):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{name}.npz")))
    X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
    eng.gp_fit(X, th, kern, m, g["f_init"])
    print("----", name, flush=True)
    r = eng.gp_fit(X, th, kern, m, g["f_init"], verbose=2)
    print(r["stats"], flush=True)
