"""Would the posterior tail of the fit (potrf of Sigma^-1 - Lambda, triangular inverse, G) overlap with mu_star, which needs
only alpha?  Sequential against two streams (two contexts, two host threads) at the C3 shape."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import Engine, get_engine
from ppbo_amd.gp_model import GPModel
from ppbo_amd.ppbo_settings import PPBO_settings
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", f"{cfg}.npz")))
D, m, th, kern = int(g["D"]), int(g["m"]), g["theta"], str(g["kernel"])
st = PPBO_settings(D=D, bounds=tuple(map(tuple, g["bounds"])), xi_acquisition_function="PCD",
                   theta_initial=list(map(float, th)), m=m, verbose=False, kernel=kern)
gp = GPModel(st)
np.random.seed(0)
gp.update_feedback_processing_object(g["X_obs"]); gp.update_data(); gp.turn_initialization_off()
gp.update_model()
eng = gp.eng
side = Engine(0)
s2 = torch.cuda.Stream(device=eng.device)
Sinv, f = gp._dSigma_inv, eng.dev(gp.fMAP)
def post_main():
    return eng.posterior(gp._dX, th, kern, Sinv, f, m)
def post_side():
    with torch.cuda.stream(s2):
        return side.posterior(gp._dX, th, kern, Sinv, f, m)
def med(fn, n=7):
    fn(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
t_post = med(post_main)
t_mu = med(lambda: gp.mu_star())
def seq():
    post_main(); gp.mu_star()
def conc():
    th_ = threading.Thread(target=post_side); th_.start(); gp.mu_star(); th_.join()
print(f"{cfg}: posterior {t_post:.2f} ms, mu_star(3) {t_mu:.2f} ms, sequential {med(seq):.2f} ms, two streams {med(conc):.2f} ms")
