"""CPU, world_size=2, gloo: the sharded-argmax exchange (the only collective on the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppbo_amd.dist import (allgather_argmax, allgather_strided, assert_same_across_ranks, broadcast_posterior, combine_best,
                           shard_bounds)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scores, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(len(scores), rank, world)
    loc = scores[lo:hi]
    if len(loc):
        li = int(np.argmax(loc))
        v, gi = float(loc[li]), lo + li
    else:
        v, gi = float("nan"), -1
    out = allgather_argmax(v, gi)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "tie_across_shards", "ragged"])
def test_sharded_argmax_two_ranks(case):
    rng = np.random.default_rng(0)
    if case == "plain":
        scores = rng.standard_normal(1000)
    elif case == "tie_across_shards":
        scores = rng.standard_normal(1000)
        scores[10] = scores[900] = 9.0       # equal maxima on both shards -> lowest index wins
    else:
        scores = rng.standard_normal(7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scores, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, (v, i) in res:
        assert i == int(np.argmax(scores))
        assert v == float(scores.max())


def test_shard_bounds_cover_everything():
    for M in (0, 1, 7, 65536, 262144):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(M, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            assert all(spans[k][1] == spans[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_combine_best_semantics():
    v, i = combine_best(torch.tensor([1.0, 3.0, 3.0, float("nan")]), torch.tensor([5, 9, 2, 1]))
    assert (v, i) == (3.0, 2)
    v, i = combine_best(torch.tensor([float("nan")]), torch.tensor([-1]))
    assert i == -1


def _worker_strided(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = [100.0 + k for k in range(rank, n_total, world)]        # item k is evaluated by rank k % world
    q.put((rank, allgather_strided(mine, n_total)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1, 7, 20, 60])
def test_theta_slices_are_gathered_in_item_order(n_total):
    """optimize_theta under --gpus N: rank r evaluates thetas[r::world]; one all-gather returns all 60 values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_strided, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, vals in res:
        assert vals == [100.0 + k for k in range(n_total)]


def test_strided_gather_single_process():
    assert allgather_strided([1.0, 2.0, 3.0], 3) == [1.0, 2.0, 3.0]


def _worker_bcast(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ppbo_amd.engine import Posterior
    N = 64
    gen = torch.Generator().manual_seed(5)
    ref = [torch.randn(N, dtype=torch.float64, generator=gen) for _ in range(3)] + [torch.randn(N, N, dtype=torch.float64, generator=gen)]
    if rank == 0:
        post = Posterior("SE_kernel", (0.1, 0.3, 0.5), 31, torch.zeros(N, 2, dtype=torch.float64), *[t.clone() for t in ref])
    else:
        post = Posterior("SE_kernel", (0.1, 0.3, 0.5), 31, torch.zeros(N, 2, dtype=torch.float64),
                         *[torch.full_like(t, float("nan")) for t in ref])
    broadcast_posterior(post, src=0)
    same = all(torch.equal(a, b) for a, b in zip((post.alpha, post.lam_diag, post.lam_off, post.G), ref))
    ok = True
    try:
        assert_same_across_ranks([1.0, 2.0], "equal values")             # must pass
    except RuntimeError:
        ok = False
    raised = False
    try:
        assert_same_across_ranks([float(rank)], "rank-dependent values")  # must raise on every rank
    except RuntimeError:
        raised = True
    q.put((rank, same, ok, raised))
    dist.destroy_process_group()


def test_model_broadcast_and_rank_consistency_check():
    """(e): the model-broadcast alternative to replicated fits, and the checksum guard evidence_batch uses."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, ok, raised in res:
        assert same and ok and raised, (rank, same, ok, raised)
