"""(e) the sharded candidate search as round 4 runs it: ppbo_predict_record (the shard's best stays on the device, the
score launch reduces its own argmax), the RCCL all-gather fed from device memory, and ppbo_search_sharded (one library
call per step).  The nccl backend and the library's own communicator are exercised at world = 1 -- this build
environment has 1-GPU boxes only -- through the same code the 8-GPU run uses.  Reference: none (the reference is
process-per-run, ppbo_numerical_main.py:192-193); what is replaced is mu_star's sequential search
(src/gp_model.py:415-437) over a sharded candidate set."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine(0)


@pytest.fixture(scope="module")
def model(eng):
    g = load_golden("c2")
    X, th, m = g["X"], g["theta"], int(g["m"])
    Sinv = eng.pd_inverse(eng.gram(eng.dev(X), th, str(g["kernel"])))
    post = eng.posterior(eng.dev(X), th, str(g["kernel"]), Sinv, g["fMAP"], m)
    return g, post


@pytest.mark.parametrize("M", [1, 37, 256, 257, 8192, 65536, 70001, 140000])
@pytest.mark.parametrize("score", [0, 2])
def test_predict_record_equals_predict_best(eng, model, M, score):
    """The device record = (best score, offset + FIRST index) of ppbo_predict on the same rows, for one launch, a
    ragged last block, and several 65536-candidate chunks; back-to-back launches of different grid sizes reuse the
    ticket counter."""
    g, post = model
    D = g["X"].shape[1]
    Xc = eng.dev(np.random.default_rng(M).random((M, D)))
    mustar = float(np.max(g["mu"]))
    ref = eng.predict(post, Xc, score=score, mustar=mustar, want_mu=False, want_var=False, want_score=True, want_best=True)
    sc = ref["score"].cpu().numpy()
    assert ref["best_idx"] == int(np.argmax(sc)) and ref["best_val"] == sc.max()
    for offset in (0, 123456789012):
        rec = eng.predict_record(post, Xc, score=score, mustar=mustar, index_offset=offset).cpu().numpy()
        assert rec[0] == ref["best_val"] and int(rec[1]) == ref["best_idx"] + offset


def test_predict_record_ties_and_repeats(eng, model):
    """Duplicate candidates tie exactly: the FIRST index wins (np.argmax), in every one of 50 repeats (the merge does
    not depend on which workgroup retires last)."""
    g, post = model
    D = g["X"].shape[1]
    base = np.random.default_rng(3).random((4096, D))
    Xc = eng.dev(np.concatenate([base, base, base]))
    want = None
    for _ in range(50):
        rec = eng.predict_record(post, Xc, score=0).cpu().numpy()
        if want is None:
            mu = eng.predict(post, Xc, want_var=False, want_best=False)["mu"].cpu().numpy()
            want = (mu.max(), int(np.argmax(mu)))
            assert want[1] < 4096
        assert (rec[0], int(rec[1])) == want


def test_search_sharded_without_a_communicator(eng, model):
    from ppbo_amd.dist import ShardedSearch
    g, post = model
    Xc = np.random.default_rng(5).random((5000, g["X"].shape[1]))
    ref = eng.predict(post, Xc, score=2, mustar=0.1, want_mu=False, want_var=False)
    for coll in ("torch", "capi"):
        s = ShardedSearch(eng, post, Xc, 1000, 2, 0.1, collective=coll)
        assert s.step() == (ref["best_val"], ref["best_idx"] + 1000)
        s.close()


def test_emulated_shards_agree_with_the_unsharded_search(eng, model):
    """Eight shards scored one after the other into a [8, 2] device record table + ppbo_argmax_combine = the argmax of
    the whole set (what the all-gather assembles on every rank)."""
    from ppbo_amd.dist import shard_bounds
    g, post = model
    M, W = 16384, 8
    Xc = eng.dev(np.random.default_rng(9).random((M, g["X"].shape[1])))
    ref = eng.predict(post, Xc, score=2, mustar=float(np.max(g["mu"])), want_mu=False, want_var=False)
    table = eng.empty(2 * W)
    for r in range(W):
        lo, hi = shard_bounds(M, r, W)
        eng.predict_record(post, Xc[lo:hi], score=2, mustar=float(np.max(g["mu"])), index_offset=lo,
                           out=table[2 * r:2 * r + 2])
    assert eng.argmax_combine(table) == (ref["best_val"], ref["best_idx"])


def test_library_communicator_at_world_1(eng, model):
    """ppbo_dist_unique_id -> ppbo_dist_init -> ppbo_argmax_allgather_record / ppbo_search_sharded with a real
    ncclUniqueId: RCCL's communicator and its all-gather kernel run (one rank)."""
    from ppbo_amd.engine import Engine
    g, post = model
    e2 = Engine(0)                       # a ctx of its own: the communicator belongs to the ctx
    post2 = post
    e2.dist_init(e2.dist_unique_id(), 0, 1)
    try:
        Xc = np.random.default_rng(6).random((3000, g["X"].shape[1]))
        ref = eng.predict(post, Xc, score=2, mustar=0.2, want_mu=False, want_var=False)
        assert e2.search_sharded(post2, Xc, 2, 0.2, 77) == (ref["best_val"], ref["best_idx"] + 77)
        rec = e2.predict_record(post2, Xc, 2, 0.2, 5)
        assert e2.argmax_allgather_record(rec) == (ref["best_val"], ref["best_idx"] + 5)
        assert e2.argmax_allgather(1.25, 9) == (1.25, 9)
    finally:
        e2.dist_destroy()


def test_torch_free_host_runs_unique_id_then_init(tmp_path):
    """in a host WITHOUT torch nothing else holds librccl between ppbo_dist_unique_id and ppbo_dist_init;
    the library must keep it mapped (RTLD_NODELETE).  tests/c/dist_smoke.c does the sequence twice on one ctx."""
    exe = tmp_path / "dist_smoke"
    subprocess.run(["gcc", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c", "dist_smoke.c"), "-ldl"], check=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([str(exe), os.path.join(ROOT, "ppbo_amd", "libppbo_hip.so")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dist_smoke ok" in r.stdout


@pytest.mark.parametrize("collective", ["torch", "capi"])
def test_bench_runs_the_nccl_path_at_world_1(collective):
    """bench.py --force-dist: init_process_group("nccl") with one rank, all_gather_into_tensor on device tensors (or
    the library's communicator), the device-side reduction -- the step the 8-GPU run times -- and the measured
    round trip of the collective in the line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--collective", collective,
                        "--config", "c2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["M_total"] == 16384
    assert "RCCL" in line["config"]["collective"]
    assert 0 < line["collective_roundtrip_us"] < 5000
    assert 0 <= line["best"]["index"] < 16384
