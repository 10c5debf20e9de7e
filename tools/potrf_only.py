import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(N)
Q = rng.standard_normal((N, N))
A = eng.dev(Q @ Q.T + N * np.eye(N))
for _ in range(10):
    eng.potrf_(A.clone())
for _ in range(3):
    eng.pd_inverse(A)
torch.cuda.synchronize()
