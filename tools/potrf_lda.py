"""Wall time of ppbo_potrf against the leading dimension (power-of-two row pitch vs padded)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
for N in (2048, 4096):
    rng = np.random.default_rng(N)
    Q = rng.standard_normal((N, N))
    A = eng.dev(Q @ Q.T + N * np.eye(N))
    for pad in (0, 8, 16, 32, 64):
        buf = eng.empty(N, N + pad)
        view = buf[:, :N]
        ts = []
        for rep in range(6):
            view.copy_(A)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.potrf_(view)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"N={N} lda={N + pad}: potrf {min(ts[1:]) * 1e3:.3f} ms")
