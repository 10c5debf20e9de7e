"""Predicted 1/2/4/8-GPU numbers for `bench.py --gpus G` (C3: N = 2048, D = 20, M_total = 65536 candidates, STRONG
scaling -- BASELINE's metric is quoted at a fixed M), from ONE GPU: every rank's leg is run here (the model is
replicated and the candidates are independent, so a rank's step time does not depend on the others), with the path's
one collective EXECUTED at world = 1 on both bindings -- torch.distributed's nccl backend (all_gather_into_tensor on the
device record + ppbo_argmax_combine) and the library's own RCCL communicator (ppbo_search_sharded) -- so that the
fixed per-step cost in the table is measured, not assumed.  What a 1-GPU box cannot measure is the extra latency of a
16-byte all-gather between 2/4/8 ranks over xGMI: it enters as an explicit allowance.

    python tools/predict_scaling.py > profiles/r04_scaling_prediction.txt
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import socket
with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
import numpy as np, torch
import torch.distributed as dist
from ppbo_amd.engine import Engine, get_engine, SCORE_POINTWISE_EI
from ppbo_amd.dist import ShardedSearch, shard_bounds

eng = get_engine(0)
XGMI_ALLOWANCE_US = (10.0, 40.0)     # extra latency of the 16-byte all-gather between G > 1 ranks (not measurable here)


def fitted(cfg):
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"{cfg}.npz")))
    X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
    S = eng.gram(X, th, kern)
    Sinv, L = eng.pd_inverse_chol(S)
    f, _ = eng.fit_fmap(Sinv, g["f_init"], m, th[0], L=L)
    return g, eng.posterior(X, th, kern, Sinv, f, m)


def step_ms(search, steps=40):
    for _ in range(5):
        search.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        search.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def legs(cfg, M_total, collective, e=None, rounds=3):
    """slowest rank's step time per G (best of `rounds` passes over all legs: the box's clocks wander by ~1 %)"""
    g, post = fitted(cfg)
    D = g["X"].shape[1]
    mustar = float(np.max(g["mu"]))
    Xall = np.random.default_rng(1).random((M_total, D))
    out = {}
    for G in (1, 2, 4, 8):
        worst = 0.0
        for r in range(G):
            lo, hi = shard_bounds(M_total, r, G)
            s = ShardedSearch(e or eng, post, Xall[lo:hi], lo, SCORE_POINTWISE_EI, mustar, collective=collective)
            s._own_comm = False          # the communicator (if any) is set up once by the caller
            worst = max(worst, min(step_ms(s) for _ in range(rounds)))
        out[G] = worst
    return out


def table(name, M_total, t, fixed_note):
    print(f"\n# {name}: {M_total} candidates in total, STRONG scaling; {fixed_note}")
    print("# G   M/rank   step ms (slowest rank, measured)   evals/s (all ranks)                 efficiency vs G = 1")
    base = M_total / (t[1] * 1e-3)
    for G in (1, 2, 4, 8):
        if G == 1:
            print(f"  {G}   {M_total // G:6d}   {t[G]:8.3f}                           {base:.3e}                           1.00")
            continue
        lo = M_total / ((t[G] + XGMI_ALLOWANCE_US[1] * 1e-3) * 1e-3)
        hi = M_total / ((t[G] + XGMI_ALLOWANCE_US[0] * 1e-3) * 1e-3)
        print(f"  {G}   {M_total // G:6d}   {t[G]:8.3f}                           {lo:.3e} - {hi:.3e}            "
              f"{lo / base / G:.3f} - {hi / base / G:.3f}   (no xGMI allowance: {M_total / (t[G] * 1e-3) / base / G:.3f})")


print("# tools/predict_scaling.py on one MI355X; every step = kstar + quadform + score(+argmax) launches, the collective, "
      "one 16-byte read-back, ONE host wait")
print(f"# allowance for the multi-rank xGMI latency of the 16-byte all-gather (not measurable on a 1-GPU box): "
      f"{XGMI_ALLOWANCE_US[0]:.0f}-{XGMI_ALLOWANCE_US[1]:.0f} us per step for G > 1")

# (a) no collective at all: the kernels + read-back
t_none = legs("c3", 65536, "capi")
# (b) the library's own communicator at world = 1 (ncclAllGather kernel + reduction on the same stream)
e2 = Engine(0)
e2.dist_init(e2.dist_unique_id(), 0, 1)
t_capi = legs("c3", 65536, "capi", e2)
# (c) torch.distributed nccl at world = 1
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t_torch = legs("c3", 65536, "torch")

print("\n# fixed cost of the collective at world = 1, per step (difference of the measured legs, ms):")
for G in (1, 2, 4, 8):
    print(f"#   M/rank {65536 // G:6d}: no collective {t_none[G]:.3f} | library communicator (ppbo_search_sharded) {t_capi[G]:.3f} "
          f"({(t_capi[G] - t_none[G]) * 1e3:+.1f} us) | torch.distributed nccl + ppbo_argmax_combine {t_torch[G]:.3f} "
          f"({(t_torch[G] - t_none[G]) * 1e3:+.1f} us)")
table("C3 (N=2048, D=20), torch.distributed nccl binding [bench.py's default]", 65536, t_torch,
      "G = 1 row of bench.py itself runs without a collective")
table("C3 (N=2048, D=20), library communicator binding [bench.py --collective capi]", 65536, t_capi, "")

# the collective alone, back to back (device record resident): all-gather + reduction + read-back + host wait
g, post = fitted("c2")
rec = eng.predict_record(post, np.random.default_rng(0).random((256, g["X"].shape[1])), SCORE_POINTWISE_EI, 0.0, 0)
gath = eng.empty(2)
for _ in range(10):
    dist.all_gather_into_tensor(gath, rec); eng.argmax_combine(gath)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    dist.all_gather_into_tensor(gath, rec); eng.argmax_combine(gath)
torch.cuda.synchronize()
print(f"\n# collective alone, torch nccl world = 1: all_gather_into_tensor + ppbo_argmax_combine (sync): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
rec2 = e2.predict_record(post, np.random.default_rng(0).random((256, g['X'].shape[1])), SCORE_POINTWISE_EI, 0.0, 0)
for _ in range(10):
    e2.argmax_allgather_record(rec2)
t0 = time.perf_counter()
for _ in range(500):
    e2.argmax_allgather_record(rec2)
print(f"# collective alone, library communicator world = 1: ppbo_argmax_allgather_record (sync): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
for _ in range(10):
    eng.argmax_combine(gath)
t0 = time.perf_counter()
for _ in range(500):
    eng.argmax_combine(gath)
print(f"# reduction + 16-byte read-back alone (ppbo_argmax_combine, sync): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
e2.dist_destroy()
dist.destroy_process_group()
