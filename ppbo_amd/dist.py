"""Multi-GPU candidate search: one process per GPU, candidates sharded by contiguous row
blocks, model state replicated (every rank fits the same deterministic model, so there is
no data-path collective), and ONE collective per search: an all-gather of the 16-byte
(best score, global index) record, combined with np.argmax tie-breaking (lowest index).

The reference has no distributed code; this replaces the sequential differential-evolution
search of mu_star (gp_model.py:415-437) for sharded candidate sets.  Backend "nccl" is RCCL
on ROCm (xGMI); "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(M: int, rank: int, world: int):
    """Contiguous row block [lo, hi) of rank `rank` (first M % world ranks get one extra row)."""
    base, rem = divmod(M, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def combine_best(vals: torch.Tensor, idxs: torch.Tensor):
    """Max value, ties -> smallest global index; entries with idx < 0 are empty."""
    best_v, best_i = None, -1
    for v, i in zip(vals.tolist(), idxs.tolist()):
        i = int(i)
        if i < 0 or v != v:
            continue
        if best_i < 0 or v > best_v or (v == best_v and i < best_i):
            best_v, best_i = v, i
    return (best_v if best_i >= 0 else float("nan")), best_i


def allgather_argmax(local_val: float, local_global_idx: int, device=None, group=None, engine=None):
    """All ranks get the global (value, index) from ONE all-gather of a 16-byte record per rank.
    The index travels as a float64 (exact below 2**53).  With an `engine` and device tensors (the RCCL path) the
    gathered records are reduced on the device (ppbo_argmax_combine: one wavefront) and one 16-byte record is
    read back; on host tensors (gloo, the CPU tests) combine_best applies the same rule."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(local_val), int(local_global_idx)
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    rec = torch.tensor([float(local_val), float(local_global_idx)], dtype=torch.float64, device=dev)
    out = torch.empty(2 * world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, rec, group=group)
    if engine is not None and out.is_cuda:
        return engine.argmax_combine(out)
    out = out.cpu().view(world, 2)
    return combine_best(out[:, 0], out[:, 1].to(torch.int64))


def sharded_search(engine, post, Xc_shard, shard_offset: int, score, mustar=0.0, group=None):
    """Score this rank's candidate rows on its GPU, then one all-gather for the argmax."""
    out = engine.predict(post, Xc_shard, score=score, mustar=mustar, want_mu=False, want_var=False,
                         want_score=False, want_best=True)
    gidx = out["best_idx"] + shard_offset if out["best_idx"] >= 0 else -1
    return allgather_argmax(out["best_val"], gidx, device=engine.device, group=group, engine=engine)


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def allgather_strided(local_vals, n_total: int, device=None, group=None):
    """Rank r holds the values of items r, r + world, r + 2 world, ... (in that order); every rank gets all
    n_total values in item order from ONE all-gather of equally padded float64 records."""
    rank, world = rank_world(group)
    if world == 1:
        assert len(local_vals) == n_total
        return [float(v) for v in local_vals]
    per = (n_total + world - 1) // world
    dev = device if device is not None else torch.device("cpu")
    rec = torch.full((per,), float("nan"), dtype=torch.float64)
    rec[:len(local_vals)] = torch.as_tensor(list(local_vals), dtype=torch.float64)
    rec = rec.to(dev)
    out = torch.empty(per * world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, rec, group=group)
    out = out.cpu().view(world, per)
    return [float(out[k % world, k // world]) for k in range(n_total)]


def assert_same_across_ranks(values, what: str, device=None, group=None):
    """Raise on every rank if `values` (a short list of floats, e.g. a checksum of inputs every rank is supposed to
    have generated identically from the same seed) differ between ranks; one all-gather."""
    rank, world = rank_world(group)
    if world == 1:
        return
    dev = device if device is not None else torch.device("cpu")
    rec = torch.as_tensor([float(v) for v in values], dtype=torch.float64).to(dev)
    out = torch.empty(world * rec.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, rec, group=group)
    out = out.cpu().view(world, -1)
    if not bool((out == out[0:1]).all()):
        raise RuntimeError(f"{what} differ between ranks (rank {rank} sees {out.tolist()}): every rank must draw them "
                           "from an identically seeded NumPy stream, or receive them from rank 0")


def broadcast_posterior(post, src: int = 0, group=None):
    """SURVEY 2.1 C2 / 8(e), the alternative to replicated fits: the rank `src` has fitted the model, every other rank
    holds a Posterior of the same shapes (e.g. from torch.empty_like) and receives alpha, Lambda_MAP (star form) and G
    by ONE broadcast each (RCCL over xGMI: 8 N^2 bytes for G, 33.5 MB at N = 2048).  In place; returns post."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return post
    for t in (post.alpha, post.lam_diag, post.lam_off, post.G):
        if t is not None:
            dist.broadcast(t, src=src, group=group)
    return post
