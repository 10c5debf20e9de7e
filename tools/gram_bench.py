"""Sustained time of ppbo_gram (SE kernel, D = 20) at N = 2048 / 4096 / 8192: back-to-back launches between two
events, once plain and once queued behind a long-running blocker so that the host's launch rate is not measured."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
th = [0.09, 0.3, 0.5]
blk = torch.randn(8192, 8192, device=eng.device)
for N in (2048, 4096, 8192):
    X = eng.dev(np.random.default_rng(7).random((N, 20)))
    out = eng.empty(N, N)
    for _ in range(3): eng.gram(X, th, out=out)
    for mode in ("plain", "queued"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if mode == "queued":
            for _ in range(4): torch.mm(blk, blk)
        e0.record()
        for _ in range(50): eng.gram(X, th, out=out)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        gb = 8.0 * N * N + 8.0 * N * 20
        print(f"gram N={N} {mode:7s}: {us:8.2f} us  {gb / us / 1e3:7.1f} GB/s  {gb / us / 1e3 / 8000 * 100:5.1f} % of 8 TB/s  variant={os.environ.get('PPBO_GRAM_VARIANT')}")
