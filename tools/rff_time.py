"""Sustained time of ppbo_rff_project (F = 4096, N = 2048, D = 20): back-to-back launches between two events, queued
behind a long-running blocker kernel so that the host's launch rate cannot be what is measured."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
rng = np.random.default_rng(3)
N, D, F = 2048, 20, 4096
X = eng.dev(rng.random((N, D))); W = eng.dev(rng.standard_normal((F, D)) / 0.3); b = eng.dev(rng.uniform(0, 2 * np.pi, F))
out = eng.empty(F, N)
blk = torch.randn(8192, 8192, device=eng.device)
for _ in range(3): eng.rff_project(X, W, b, 0.5, out=out)
for mode in ("plain", "queued behind a blocker"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if mode != "plain":
        for _ in range(4): torch.mm(blk, blk)          # ~ms of GPU work: the launches below queue up behind it
    e0.record()
    for _ in range(50): eng.rff_project(X, W, b, 0.5, out=out)
    e1.record(); e1.synchronize()
    print(f"rff_project back-to-back avg us ({mode}): {e0.elapsed_time(e1) / 50 * 1e3:.2f}  PPBO_RFF_NT={os.environ.get('PPBO_RFF_NT')}")
