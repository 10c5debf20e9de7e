"""Does the Gram kernel's sustained rate depend on what else the process has allocated?  (bench.py saw 38 us at
N = 4096 where tools/gram_bench.py sees 24 us.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_POINTWISE_EI
eng = get_engine(0)
th = [0.09, 0.3, 0.5]
g = dict(np.load("tests/golden/c3.npz"))
blk = torch.randn(8192, 8192, device=eng.device)

def measure(tag, N=4096, out=None):
    X = eng.dev(np.random.default_rng(7).random((N, 20)))
    out = eng.empty(N, N) if out is None else out
    for _ in range(3): eng.gram(X, th, out=out)
    torch.cuda.synchronize()
    for _ in range(4): torch.mm(blk, blk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): eng.gram(X, th, out=out)
    e1.record(); e1.synchronize()
    print(f"{tag:50s} gram N={N}: {e0.elapsed_time(e1) / 40 * 1e3:7.2f} us   out ptr {out.data_ptr():#x}")
    return out

o = measure("fresh process")
Xd = eng.dev(g["X"])
S = eng.gram(Xd, g["theta"]); Sinv = eng.pd_inverse(S)
post = eng.posterior(Xd, g["theta"], "SE_kernel", Sinv, g["fMAP"], int(g["m"]))
Xc = eng.dev(np.random.default_rng(1).random((65536, 20)))
eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1)
measure("after a 65536-candidate predict (1 GB workspace)", out=o)
measure("same, new output buffer")
big = [eng.empty(2048, 4096) for _ in range(8)]
del big
measure("after torch allocator churn, new buffer")
# blocker = predict instead of torch.mm
X = eng.dev(np.random.default_rng(7).random((4096, 20)))
out = eng.empty(4096, 4096)
for _ in range(3): eng.gram(X, th, out=out)
torch.cuda.synchronize()
eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1, want_mu=False, want_var=False, want_best=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): eng.gram(X, th, out=out)
e1.record(); e1.synchronize()
print(f"blocker = predict: {e0.elapsed_time(e1) / 40 * 1e3:7.2f} us")

# ---- does half a second of fp64-MFMA work (the benchmark's timed loop) slow the next memory-bound kernel down?
import time
def burst(N=4096):
    X = eng.dev(np.random.default_rng(7).random((N, 20)))
    o = eng.empty(N, N)
    for _ in range(3): eng.gram(X, th, out=o)
    torch.cuda.synchronize()
    eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1, want_mu=False, want_var=False, want_best=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): eng.gram(X, th, out=o)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 40 * 1e3
torch.cuda.synchronize(); time.sleep(1.0)
print("idle 1 s, then burst:", [round(burst(), 2) for _ in range(3)])
for _ in range(100): eng.predict(post, Xc, score=SCORE_POINTWISE_EI, mustar=0.1)
print("after 100 scoring steps:", [round(burst(), 2) for _ in range(3)])
print("N=2048:", [round(burst(2048), 2) for _ in range(3)], " N=8192:", [round(burst(8192), 2) for _ in range(2)])
