"""Round 6: start delay of the second dispatch round of the one-launch scoring kernel (PPBO_FUSED_DBG bits 4..7)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ppbo_amd.engine import Engine  # noqa: E402
from r6_fused_check import synth_post, timed  # noqa: E402

for (N, D, m, M) in [(512, 6, 31, 16384), (512, 6, 31, 65536), (416, 6, 25, 16384), (256, 6, 31, 16384)]:
    row = []
    for delay in (0, 1, 2, 3, 4, 6):
        os.environ["PPBO_FUSED"] = "1"
        os.environ["PPBO_FUSED_DBG"] = str(delay << 4)
        e = Engine(0)
        p = synth_post(e, N, D, m, "SE_kernel", (0.001, 0.26, 0.1))
        x = e.dev(np.random.default_rng(1).random((M, D)))
        e.profile(True)
        t = timed(lambda: e.predict(p, x, score=1, mustar=0.1, want_mu=False, want_var=False), 40) * 1e3
        ms, n = e.profile_read("fused_score")
        row.append(f"delay {delay}: step {t:.1f} kernel {ms / n * 1e3:.1f}")
        e.close()
    print(f"N={N} M={M}: " + " | ".join(row), flush=True)
