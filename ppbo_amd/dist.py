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


class ShardedSearch:
    """The sharded candidate search with everything that can be set up once set up once: persistent device tensors
    for this rank's 16-byte record and for the gathered records (no tensor is built from Python floats inside a
    step, no host value is uploaded), the shard's row offset, and the choice of collective:

      collective="torch"  torch.distributed.all_gather_into_tensor on the device records (backend nccl = RCCL), then
                          ppbo_argmax_combine: one wavefront reduces the records, one 16-byte read-back;
      collective="capi"   the ctx's own RCCL communicator behind the C-ABI: the whole step is ONE library call
                          (ppbo_search_sharded: scoring, ncclAllGather, reduction, read-back on one stream, one host
                          wait).  The 128-byte ncclUniqueId is created on rank 0 and broadcast through
                          torch.distributed's store-backed object broadcast (any transport would do).

    Per step the host waits exactly once, for the reduced record.  On host tensors (gloo: the CPU tests and the
    shared-GPU test mode) the record is copied to the host and combine_best applies the same rule."""

    def __init__(self, engine, post, Xc_shard, shard_offset: int, score, mustar=0.0, group=None, collective="torch",
                 host_collective=False):
        self.eng, self.post, self.Xc = engine, post, engine.dev(Xc_shard)
        self.offset, self.score, self.mustar, self.group = int(shard_offset), score, float(mustar), group
        self.rank, self.world = rank_world(group)
        self.host = bool(host_collective)
        if collective not in ("torch", "capi"):
            raise ValueError("collective must be 'torch' or 'capi'")
        self.collective = collective
        self.record = engine.empty(2)
        self.gathered = engine.empty(2 * self.world)
        # the ctypes arguments of a step are built once (the model descriptor alone costs ~10 us of Python per call)
        import ctypes as _C
        from .engine import SCORE_MEAN, _ptr
        self._md = engine._model(post, score != SCORE_MEAN)
        self._md_ref = _C.byref(self._md)
        self._xc_ptr, self._M = _ptr(self.Xc), int(self.Xc.shape[0])
        self._bv, self._bi = _C.c_double(0.0), _C.c_int64(-1)
        self._bv_ref, self._bi_ref = _C.byref(self._bv), _C.byref(self._bi)
        self._rec_ptr = _ptr(self.record)
        if self.host:
            self.h_gathered = torch.empty(2 * self.world, dtype=torch.float64)
        if collective == "capi" and not self.host and (self.world > 1 or dist.is_initialized()):
            ids = [engine.dist_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(ids, src=0, group=group)
            engine.dist_init(ids[0], self.rank, self.world)
            self._own_comm = True
        else:
            self._own_comm = False

    def step(self):
        """-> job-wide (best score, global row index), identical on every rank."""
        eng = self.eng
        # one rank and no process group: the library call that scores and publishes the record through the ctx's
        # host-mapped block (the host polls a flag) -- ppbo_predict_record + record.tolist() paid a device-to-host copy and
        # a stream synchronisation for 16 bytes (~10 us of a 120 us step at C2)
        single = self.world == 1 and not self.host and not (dist.is_available() and dist.is_initialized())
        if (self.collective == "capi" or single) and not self.host:
            rc = eng.lib.ppbo_search_sharded(eng.ctx, self._md_ref, self._xc_ptr, self._M, int(self.score), self.mustar,
                                             self.offset, self._bv_ref, self._bi_ref, eng._stream())
            eng._check(rc, "ppbo_search_sharded")
            return self._bv.value, self._bi.value
        rc = eng.lib.ppbo_predict_record(eng.ctx, self._md_ref, self._xc_ptr, self._M, int(self.score), self.mustar,
                                         self.offset, self._rec_ptr, eng._stream())
        eng._check(rc, "ppbo_predict_record")
        if self.world == 1 and not dist.is_initialized():
            v, i = self.record.tolist()
            return v, int(i)
        if self.host:
            dist.all_gather_into_tensor(self.h_gathered, self.record.cpu(), group=self.group)
            out = self.h_gathered.view(self.world, 2)
            return combine_best(out[:, 0], out[:, 1].to(torch.int64))
        dist.all_gather_into_tensor(self.gathered, self.record, group=self.group)
        return eng.argmax_combine(self.gathered)

    def close(self):
        if self._own_comm:
            self.eng.dist_destroy()
            self._own_comm = False


def sharded_search(engine, post, Xc_shard, shard_offset: int, score, mustar=0.0, group=None):
    """Score this rank's candidate rows on its GPU, then one all-gather for the argmax (one-off form of
    ShardedSearch: builds the persistent tensors, runs one step)."""
    _, world = rank_world(group)
    host = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "gloo"
    return ShardedSearch(engine, post, Xc_shard, shard_offset, score, mustar, group, host_collective=host).step()


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
    post.Gt = None            # the cached transpose of G (engine._model) belongs to the old contents
    return post
