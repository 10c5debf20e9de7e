"""Steady-state Gram time (HIP-graph replay, streaming over rotated output buffers as bench.py does) at N = 2048 / 4096
for the tile-order variants: PPBO_GRAM_VARIANT is read per ctx, so each variant gets its own Engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import Engine
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

for rnd in range(2):
  for var in sys.argv[1:] or ["0", "2", "1", "3"]:
    os.environ["PPBO_GRAM_VARIANT"] = var
    eng = Engine(0)
    for Ng in (2048, 4096):
        Xg = eng.dev(np.random.default_rng(7).random((Ng, 20)))
        nbuf = max(1, -(-(512 * 2 ** 20) // (8 * Ng * Ng)))
        bufs = [eng.empty(Ng, Ng) for _ in range(nbuf)]
        st = {"k": 0}
        def rot():
            st["k"] = (st["k"] + 1) % nbuf
            eng.gram(Xg, th, out=bufs[st["k"]])
        res = graph_time(lambda: eng.gram(Xg, th, out=bufs[0]))
        stream = graph_time(rot)
        def fl():
            st["k"] = (st["k"] + 1) % nbuf
            eng.store_floor(bufs[st["k"]])
        floor = graph_time(fl)
        gb = 8.0 * Ng * Ng + 8.0 * Ng * 20
        print(f"variant {var} N={Ng}: resident {res:6.2f} us ({gb / res / 1e3 / 8000:.3f}) | streaming {stream:6.2f} us ({gb / stream / 1e3 / 8000:.3f} of 8 TB/s) | write-only floor {floor:6.2f} us", flush=True)
        del bufs
