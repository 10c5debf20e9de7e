"""Event-bracketed time of ppbo_rff_project (F = 4096, N = 2048, D = 20) over back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
rng = np.random.default_rng(3)
N, D, F = 2048, 20, 4096
X = eng.dev(rng.random((N, D))); W = eng.dev(rng.standard_normal((F, D)) / 0.3); b = eng.dev(rng.uniform(0, 2 * np.pi, F))
for _ in range(3): eng.rff_project(X, W, b, 0.5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.rff_project(X, W, b, 0.5)
e1.record(); e1.synchronize()
print("rff_project back-to-back avg us:", e0.elapsed_time(e1) / 20 * 1e3)
