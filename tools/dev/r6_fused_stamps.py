"""Round 6: phase stamps of the one-launch scoring kernel (PPBO_FUSED_DBG=8: the library prints the mean phase lengths of
wavefront 0 over the workgroups to stderr).  python tools/dev/r6_fused_stamps.py [N D m M]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["PPBO_FUSED"] = "2"
os.environ["PPBO_FUSED_DBG"] = "8"
from ppbo_amd.engine import Engine  # noqa: E402
from r6_fused_check import synth_post  # noqa: E402
N, D, m, M = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 6, 31, 16384)
e = Engine(0)
p = synth_post(e, N, D, m, "SE_kernel", (0.001, 0.26, 0.1))
x = e.dev(np.random.default_rng(1).random((M, D)))
for _ in range(4):
    e.predict(p, x, score=1, mustar=0.1, want_mu=False, want_var=False)
