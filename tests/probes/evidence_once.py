"""Probe: five Laplace evidences at the C3 shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
from test_gpu_golden_r2 import _fitted
class G:
    def __call__(self, n): return load_golden(n)
g, gp, st = _fitted(G(), "c3")
np.random.seed(0)
for _ in range(5): gp.evidence(list(gp.theta), None)
torch.cuda.synchronize()
