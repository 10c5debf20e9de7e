"""(e) the path's one collective, device side: the reduction of the gathered (value, global index) records
(ppbo_argmax_combine, the kernel ppbo_argmax_allgather also uses) against dist.combine_best / np.argmax semantics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine(0)


@pytest.mark.parametrize("world", [1, 2, 8, 64, 200])
def test_argmax_combine_matches_the_host_rule(eng, world):
    from ppbo_amd.dist import combine_best
    rng = np.random.default_rng(world)
    for trial in range(20):
        vals = rng.integers(0, 4, world).astype(float)          # many ties
        idx = rng.permutation(10 * world)[:world].astype(float)
        vals[rng.random(world) < 0.2] = np.nan                  # NaN scores never win
        idx[rng.random(world) < 0.2] = -1.0                     # empty shards never win
        rec = eng.dev(np.stack([vals, idx], axis=1))
        v, i = eng.argmax_combine(rec)
        ev, ei = combine_best(torch.as_tensor(vals), torch.as_tensor(idx).to(torch.int64))
        assert i == ei
        assert (v == ev) or (v != v and ev != ev)


def test_argmax_combine_first_occurrence(eng):
    rec = eng.dev(np.array([[3.0, 900.0], [3.0, 10.0], [1.0, 5.0]]))
    assert eng.argmax_combine(rec) == (3.0, 10)
    rec = eng.dev(np.array([[float("nan"), 4.0], [2.0, -1.0]]))
    v, i = eng.argmax_combine(rec)
    assert i == -1 and v != v


@pytest.mark.parametrize("N", [2, 130, 512, 1000])
def test_store_floor_probe_writes_every_entry(eng, N):
    """ppbo_store_floor (the write-only ceiling bench.py times beside ppbo_gram) covers the whole N x N matrix,
    ragged tile edges included, and rejects an odd N."""
    out = torch.full((N, N), float("nan"), dtype=torch.float64, device=eng.device)
    eng.store_floor(out)
    assert not bool(torch.isnan(out).any())
    if N > 2:
        with pytest.raises(RuntimeError):
            eng.store_floor(torch.empty((N - 1, N - 1), dtype=torch.float64, device=eng.device))
