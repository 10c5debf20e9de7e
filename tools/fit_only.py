"""One c3 GP fit (Gram, inverse, f_MAP, posterior) a few times -- target for rocprofv3 --kernel-trace."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c3.npz")))
X = eng.dev(g["X"]); m = int(g["m"]); th = g["theta"]
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = eng.gram(X, th); Sinv = eng.pd_inverse(S)
    f, st = eng.fit_fmap(Sinv, g["f_init"], m, th[0])
    post = eng.posterior(X, th, "SE_kernel", Sinv, f, m)
    torch.cuda.synchronize()
    print(f"rep {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", st)
