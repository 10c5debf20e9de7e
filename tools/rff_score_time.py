"""Kernel time of ppbo_rff_score at C3 (M = 65536 candidates, F = 4096 features, D = 20) through the library's event
brackets; PPBO_RFF_SCORE_MFMA=1 selects the matrix-core experiment."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
rng = np.random.default_rng(3)
M, D, F = 65536, 20, 4096
Xc = eng.dev(rng.random((M, D))); W = eng.dev(rng.standard_normal((F, D)) / 0.3); b = eng.dev(rng.uniform(0, 2 * np.pi, F))
om = eng.dev(rng.standard_normal(F))
for _ in range(3): eng.rff_score(Xc, W, b, 0.5, om, want_score=False)
eng.profile(True)
for _ in range(20): eng.rff_score(Xc, W, b, 0.5, om, want_score=False)
torch.cuda.synchronize()
ms, n = eng.profile_read("rff_score")
sc, bv, bi = eng.rff_score(Xc, W, b, 0.5, om)
print(f"rff_score kernel avg ms: {ms / n:.4f}  (PPBO_RFF_SCORE_MFMA={os.environ.get('PPBO_RFF_SCORE_MFMA')})  best {bv:.12f} at {bi}  checksum {float(sc.sum()):.10f}")
