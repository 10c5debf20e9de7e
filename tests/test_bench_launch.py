"""bench.py's own multi-GPU launcher: `--gpus N` must really start N ranks (or fail loudly), never run world=1
under an N-GPU label.  CPU part runs in the build container; the RCCL part needs >= 2 visible GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_must_match_world_size():
    """Under a launcher (WORLD_SIZE set) a mismatching --gpus is an error, not a silently different job."""
    r = _run(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_gpus_2_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"], timeout=300)
    assert r.returncode != 0                       # the spawned ranks refuse to run without a GPU
    assert "no ROCm GPU visible" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())    # and no JSON line pretends otherwise


@pytest.mark.gpu
def test_gpus_2_runs_two_nccl_ranks():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL path); the driver's SCALE run exercises it on the 8-GPU node")
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c2", "--scaling", "strong",
              "--no-cpu-baseline", "--no-secondary"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["M_total"] == 16384 and line["config"]["M_per_gpu"] == 8192
    assert line["replicated_fit_bitwise_equal"] is True          # two different GPUs, the same bits
    assert 0 <= line["best"]["index"] < 16384
    # the same job on one rank must pick the same candidate? no: ranks draw their own candidates (seed 1 + rank);
    # what must hold is that the winning index lies in the shard that reported it
    assert line["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_gpus_1_line_is_well_formed():
    r = _run(["--steps", "3", "--warmup", "1", "--config", "c2", "--no-cpu-baseline", "--no-secondary"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["name"] == "c2"
    assert line["config"]["M_total"] == 16384 and line["ms_per_step_with_event_brackets"] > 0
    assert 0 < line["roofline"]["frac"] <= 1.0
    # `traffic` = the counter figure of the committed PMC capture when it was taken at THIS launch shape (round 6:
    # profiles/r06_pmc_hot_kernels.json holds the C2 launch of the one-launch scoring kernel), null otherwise; either way
    # the algorithmic bytes stand beside it and the source says whether the capture belongs to the tree's kernel sources
    tr, src = line["roofline"]["traffic"], line["roofline"].get("traffic_source")
    assert line["roofline"]["algorithmic_bytes"] > 0
    assert tr is None or (tr > 0 and src["shape_matches_this_launch"] and isinstance(src["current"], bool)
                          and abs(line["roofline"]["traffic_over_algorithmic"] - tr / line["roofline"]["algorithmic_bytes"]) < 1e-9)
    assert line["roofline"]["dense_equivalent_tflops"] >= line["roofline"]["achieved"]


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_ranks_sharing_one_gpu(scaling):
    """The multi-rank control flow of bench.py on a 1-GPU box: two ranks on cuda:0, collectives over gloo
    (PPBO_BENCH_SHARE_GPU=1; RCCL refuses two ranks per device).  Sharding, barriers, the max-over-ranks timing and the
    argmax exchange are the code the 8-GPU run uses; only the backend differs."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c2", "--scaling", scaling,
              "--no-cpu-baseline", "--no-secondary"], {"PPBO_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    if scaling == "strong":
        assert line["config"]["M_total"] == 16384 and line["config"]["M_per_gpu"] == 8192
    else:
        assert line["config"]["M_total"] == 32768 and line["config"]["M_per_gpu"] == 16384
    assert 0 <= line["best"]["index"] < line["config"]["M_total"]
    assert abs(line["value"] - line["config"]["M_total"] * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-6 * line["value"]
    assert "TEST MODE" in line["config"]["collective"]
    # every rank fitted the same model from the same inputs: alpha and G agree bit for bit (the replicas-instead-of-
    # broadcast assumption of DESIGN 6)
    assert line["replicated_fit_bitwise_equal"] is True
    other = line["weak_scaling_leg" if scaling == "strong" else "strong_scaling_leg"]
    assert other["M_total"] == (32768 if scaling == "strong" else 16384) and other["value"] > 0


@pytest.mark.gpu
def test_c3_defaults_to_strong_scaling_at_the_metrics_candidate_count():
    """BASELINE's metric is quoted at a FIXED M = 65536 on 1/2/4/8 GPUs: `bench.py --gpus 2` with no other flag must
    shard those 65536 candidates (32768 per rank) and report the strong-scaling rate as `value`; the weak leg (65536
    per rank) rides along as a side key."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
             {"PPBO_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["name"] == "c3" and line["n_gpus"] == 2
    assert line["scaling"] == "strong" and line["config"]["M_total"] == 65536 and line["config"]["M_per_gpu"] == 32768
    assert "strong" in line["config"]["parallelism"]
    assert line["weak_scaling_leg"]["M_total"] == 131072 and line["weak_scaling_leg"]["M_per_gpu"] == 65536
    assert 0 <= line["best"]["index"] < 65536


@pytest.mark.gpu
def test_chunked_config_keeps_the_roofline_fraction_below_one():
    """C4 scores 262144 candidates in four 65536-candidate launches per step: the per-launch flops must be priced per
    launch (a first version priced the whole step against one launch's time: frac 3.2)."""
    r = _run(["--steps", "2", "--warmup", "1", "--config", "c4", "--no-cpu-baseline", "--no-secondary"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["M_total"] == 262144
    assert 0 < line["roofline"]["frac"] <= 1.0
    assert line["roofline"]["launches"] == 4 * 2
