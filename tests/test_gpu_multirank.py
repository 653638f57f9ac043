"""
The REAL (non --dry-run) multi-rank path of bench.py on one MI355X: `python bench.py --gpus 2` launches
its two ranks itself, both use cuda:0 (`--share-gpu`), the process group is gloo (RCCL refuses two ranks on one device;
the gathered rows are staged through the host there, neo_planner_amd/sharding.py) -- everything else is the code the
driver's 2/4/8-GPU runs execute: per-rank scene and request batches, the barrier-fenced timed region, the gather of
every batch's results inside it, max-over-ranks time, one JSON line from rank 0.  The children are started before
anything touches the GPU (bench.self_launch); no process that has initialised HIP is ever re-executed.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _bench(args, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_real_path():
    out = _bench(["--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu",
                  "--batches-per-step", "2", "--streams", "2"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dist_backend"] == "gloo"
    assert out["scaling"] == "weak" and out["steps"] == 2
    assert out["value"] is not None and out["value"] > 1e4 and out["value"] == out["value"]
    assert len(out["per_rank_traj_per_s"]) == 2 and all(v > 0 for v in out["per_rank_traj_per_s"])
    assert out["gather_ok"] is True
    # whole-job value = all ranks' trajectories over the slowest rank's time
    assert out["value"] <= 1.001 * sum(out["per_rank_traj_per_s"])
    assert out["config"]["parallelism"] == "scene-sharded x2"


def test_cfg4_default_scene_count_matches_baseline_config():
    """256 scenes over the 8 GPUs of a node = 32 per GPU (BASELINE.json configs[3]); here a short run on one GPU"""
    out = _bench(["--config", "cfg4", "--steps", "1", "--warmup", "1", "--no-cpu"])
    assert out["n_gpus"] == 1 and "32 x 300^3" in out["config"]["workload"]
    assert out["config"]["batch_per_launch"] == 32 * 4096 and out["value"] > 1e4


def test_rccl_gather_behind_each_batch_with_one_rank():
    """the process-group path on RCCL itself (backend "nccl"), as far as one GPU allows -- a one-rank
    group (NEO_BENCH_FORCE_DIST=1): every batch's packed results go through an asynchronous all_gather_into_tensor on the
    process group's stream behind the batch, the gathered rows are the rank's rows bit for bit, and the collective costs
    the hot path next to nothing (it is off the critical path: DESIGN.md section 6)."""
    args = ["--steps", "4", "--warmup", "2", "--no-cpu", "--no-report", "--no-modes"]
    plain = _bench(args)
    e = dict(os.environ)
    os.environ["NEO_BENCH_FORCE_DIST"] = "1"
    try:
        dist = _bench(args + ["--dist-backend", "nccl"])
    finally:
        os.environ.clear()
        os.environ.update(e)
    assert "rccl_ranks" not in plain and "gather_ok" not in plain          # (no process group in the plain run)
    assert dist["rccl_ranks"] == 1 and dist["dist_backend"] == "nccl" and dist["gather_ok"] is True
    assert dist["n_gpus"] == 1 and dist["steps"] == 4
    assert dist["value"] >= 0.95 * plain["value"], (dist["value"], plain["value"])
