"""Round 6: kernel time of the one-launch scoring kernel against the number of workgroups per CU (N = 512, D = 6)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ppbo_amd.engine import Engine  # noqa: E402
from r6_fused_check import synth_post, timed  # noqa: E402
e = Engine(0)
N, D, m = 512, 6, 31
p = synth_post(e, N, D, m, "SE_kernel", (0.001, 0.26, 0.1))
for M in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 65536, 131072):
    x = e.dev(np.random.default_rng(1).random((M, D)))
    e.profile(True)
    timed(lambda: e.predict(p, x, score=1, mustar=0.1, want_mu=False, want_var=False), 30)
    ms, n = e.profile_read("fused_score")
    e.profile(False)
    k = ms / n * 1e3
    mf = 2.0 * 16 * M * sum(min(N, ((16 * s + 16 + 31) // 32) * 32) for s in range(N // 16)) / 1e9
    print(f"M={M:6d}: {M // 32:5d} workgroups ({M / 32 / 256:.2f} per CU)  kernel {k:7.1f} us  {mf / k * 1e-3:.1f} TFLOP/s executed", flush=True)
