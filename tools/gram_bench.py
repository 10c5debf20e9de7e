"""Time the Gram kernel (event timing inside the library) at several N; prints GB/s vs 8 TB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
for N, D in [(2048, 20), (4096, 6), (4096, 20), (8192, 20)]:
    X = eng.dev(np.random.default_rng(0).random((N, D)))
    for _ in range(3):
        eng.gram(X, [0.09, 0.3, 0.5])
    eng.profile(True)
    for _ in range(20):
        eng.gram(X, [0.09, 0.3, 0.5])
    torch.cuda.synchronize()
    ms, n = eng.profile_read("gram")
    eng.profile(False)
    by = 8.0 * N * N + 8.0 * N * D
    print(f"gram N={N} D={D}: {ms/n*1e3:.1f} us  {by/(ms/n*1e-3)/1e9:.0f} GB/s  frac {by/(ms/n*1e-3)/1e9/8000:.3f}")
