import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_MEAN, SCORE_POINTWISE_EI
eng = get_engine(0)
g = dict(np.load("tests/golden/c3.npz"))
X, th, m = g["X"], g["theta"], int(g["m"])
Sinv = eng.pd_inverse(eng.gram(X, th))
post = eng.posterior(X, th, "SE_kernel", Sinv, g["fMAP"], m)
Xc = eng.dev(np.random.default_rng(1).random((65536, 20)))
for name, kw in (("mean only (no K* store)", dict(score=SCORE_MEAN, want_var=False, want_mu=False)),
                 ("full", dict(score=SCORE_POINTWISE_EI, mustar=0.1, want_var=False, want_mu=False))):
    for _ in range(3): eng.predict(post, Xc, **kw)
    eng.profile(True)
    for _ in range(10): eng.predict(post, Xc, **kw)
    torch.cuda.synchronize()
    ms, n = eng.profile_read("kstar")
    print(name, "kstar avg ms", ms / n)
    eng.profile(False)
