"""Wall time of the device factorizations (potrf, pd_inverse) and the c3 fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
for N in (512, 1024, 2048, 4096):
    rng = np.random.default_rng(N)
    Q = rng.standard_normal((N, N))
    A = eng.dev(Q @ Q.T + N * np.eye(N))
    for name, fn in (("potrf", lambda: eng.potrf_(A.clone())), ("pd_inverse", lambda: eng.pd_inverse(A))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(f"{name} N={N}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
g = dict(np.load("tests/golden/c3.npz"))
S = eng.gram(g["X"], g["theta"]); Sinv = eng.pd_inverse(S)
for gtol in (1e-4, 1e-6):
    eng.fit_fmap(Sinv, g["f_init"], int(g["m"]), g["theta"][0], gtol=gtol)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f, st = eng.fit_fmap(Sinv, g["f_init"], int(g["m"]), g["theta"][0], gtol=gtol)
    torch.cuda.synchronize()
    print(f"fit c3 gtol={gtol}: {(time.perf_counter()-t0)*1e3:.1f} ms", st)
