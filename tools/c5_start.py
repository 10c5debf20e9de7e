"""GPU box: fit the reference-built C5 design on the device and save the result as a warm start for the
reference's own trust-exact run (which takes hours from a cold start at N=4096)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from ppbo_amd.engine import get_engine
eng = get_engine(0)
g = dict(np.load("tests/golden/_c5_design.npz"))
X, th, m, kern = g["X"], g["theta"], int(g["m"]), str(g["kernel"])
N = X.shape[0]
S = eng.gram(X, th, kern)
Sinv = eng.pd_inverse(S)
f0 = eng.dgemv(eng.potrf_(S.clone()), np.random.default_rng(2).standard_normal(N), lower=True)
t0 = time.time()
f, st = eng.fit_fmap(Sinv, f0, m, th[0], gtol=1e-7)
print("fit", st, "%.1fs" % (time.time() - t0))
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/c5_start.npy", f.cpu().numpy())
