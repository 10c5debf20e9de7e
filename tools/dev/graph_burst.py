"""Steady-state time of a short kernel: 100 launches captured in a HIP graph (torch.cuda.CUDAGraph over the stream the
C-ABI launches on), replayed back to back for tens of milliseconds so that the clocks settle under the kernel's OWN
load; the last replays are timed.  Compared with the spin-kernel-blocker burst and the torch.mm-blocker burst."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
N, D, F = 2048, 20, 4096
rng = np.random.default_rng(3)
X = eng.dev(rng.random((N, D)))
W = eng.dev(rng.standard_normal((F, D)) / 0.3); b = eng.dev(rng.uniform(0, 2 * np.pi, F))
th = [0.09, 0.3, 0.5]

def graph_time(fn, per_graph=100, replays=40, timed=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(per_graph): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(replays):
        if r == replays - timed: e0.record()
        gr.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (timed * per_graph) * 1e3

out = eng.empty(F, N)
print("rff_project F=4096 N=2048: graph steady state %.2f us" % graph_time(lambda: eng.rff_project(X, W, b, 0.5, out=out)))
for Ng in (2048, 4096, 8192):
    Xg = eng.dev(np.random.default_rng(7).random((Ng, 20))); o = eng.empty(Ng, Ng)
    us = graph_time(lambda: eng.gram(Xg, th, out=o), per_graph=50 if Ng < 8192 else 20)
    gb = 8.0 * Ng * Ng + 8.0 * Ng * 20
    print("gram N=%d: graph steady state %.2f us = %.3f of 8 TB/s" % (Ng, us, gb / us / 1e3 / 8000))
