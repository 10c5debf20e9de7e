"""Why does bench.py see 17.3 us for rff_project where tools/rff_time.py sees 15.6 us?  Same measurement, varying what
ran before and which inputs are used."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
g = dict(np.load("tests/golden/c3.npz"))
N, D, F = 2048, 20, 4096
rng = np.random.default_rng(3)
Xr = eng.dev(rng.random((N, D))); Xf = eng.dev(g["X"])
W = eng.dev(np.random.default_rng(3).standard_normal((F, D)) / 0.3); b = eng.dev(np.random.default_rng(4).uniform(0, 2 * np.pi, F))

def burst(X, tag):
    out = eng.empty(F, N)
    for _ in range(3): eng.rff_project(X, W, b, 0.5, out=out)
    torch.cuda.synchronize()
    torch.cuda._sleep(40_000_000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): eng.rff_project(X, W, b, 0.5, out=out)
    e1.record(); e1.synchronize()
    print(f"{tag:45s} {e0.elapsed_time(e1) / 40 * 1e3:6.2f} us")

burst(Xr, "fresh, random X")
burst(Xf, "fresh, fixture X")
Sig = eng.gram(Xf, g["theta"]); Sinv = eng.pd_inverse(Sig)
f, st = eng.fit_fmap(Sinv, g["f_init"], int(g["m"]), g["theta"][0])
post = eng.posterior(Xf, g["theta"], "SE_kernel", Sinv, f, int(g["m"]))
torch.cuda.synchronize()
burst(Xr, "after a fit, random X")
burst(Xf, "after a fit, fixture X")
Xc = eng.dev(np.random.default_rng(1).random((65536, D)))
eng.predict(post, Xc)
burst(Xr, "after a predict (1 GB workspace), random X")
burst(Xf, "after a predict, fixture X")
