"""Round 6: where the one-launch scoring kernel's time goes (PPBO_FUSED_DBG: 1 = no kernel evaluations, 2 = no contraction)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ppbo_amd.engine import Engine  # noqa: E402
from r6_fused_check import synth_post, timed  # noqa: E402


def main():
    engs = {}
    for dbg in (0, 1, 2, 3, 5):
        os.environ["PPBO_FUSED"] = "1"
        os.environ["PPBO_FUSED_DBG"] = str(dbg)
        engs[dbg] = Engine(0)
    import itertools
    for (N, D, m, M), kern in itertools.product([(512, 6, 31, 16384), (512, 20, 31, 16384), (1024, 10, 31, 65536), (256, 6, 31, 16384)],
                                                 ("SE_kernel", "RQ_kernel")):
        row = []
        for dbg, e in engs.items():
            p = synth_post(e, N, D, m, kern, (0.001, 0.26, 0.1))
            x = e.dev(np.random.default_rng(1).random((M, D)))
            row.append(timed(lambda: e.predict(p, x, score=1, mustar=0.1, want_mu=False, want_var=False), 30) * 1e3)
        print(f"{kern[:2]} N={N} D={D} M={M}: full {row[0]:.1f} us | no evals {row[1]:.1f} | no contraction {row[2]:.1f} | neither {row[3]:.1f} | no evals, no G loads in the loop {row[4]:.1f}", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
