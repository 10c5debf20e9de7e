"""Round 6: N launches of the scoring step at a synthetic model, for rocprofv3 (--kernel-trace / --pmc).
   python3 tools/dev/r6_fused_run.py N D m M reps [fused=1]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
N, D, m, M, reps = (int(v) for v in sys.argv[1:6])
os.environ["PPBO_FUSED"] = sys.argv[6] if len(sys.argv) > 6 else "1"
from ppbo_amd.engine import Engine  # noqa: E402
from r6_fused_check import synth_post  # noqa: E402
e = Engine(0)
p = synth_post(e, N, D, m, "SE_kernel", (0.001, 0.26, 0.1))
x = e.dev(np.random.default_rng(1).random((M, D)))
for _ in range(reps):
    e.predict(p, x, score=1, mustar=0.1, want_mu=False, want_var=False)
