"""CPU tests of bench.py's multi-rank protocol: `python bench.py --gpus 2` launches its two ranks itself
(no torchrun), they rendezvous on 127.0.0.1 (gloo here, RCCL on GPUs), run the barrier-fenced timed loop with the
per-step gather of packed results, and rank 0 prints ONE JSON line.  The device work is stubbed (--dry-run):
no throughput is claimed, the plumbing is what is under test.  Also: the usable-CPU count of the CPU baseline."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_end_to_end():
    out = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["gather_ok"] is True
    assert out["steps"] == 4 and out["value"] is None and out["dry_run"] is True


def test_self_launch_eight_ranks_cfg4_scene_split():
    """the 8-rank launch of BASELINE configs[3] as far as CPUs allow -- eight processes rendezvous (gloo),
    each holds 256 / 8 = 32 scenes, and the rank-major gather of their result rows comes back scene-major"""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--config", "cfg4"])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["gather_ok"] is True and out["dry_run"] is True
    assert out["scenes_per_rank"] == 32 and out["scenes_total"] == 256 and out["scene_major_ok"] is True


def test_torchrun_style_environment_is_respected():
    """when the driver launches the ranks (python -m torch.distributed.run ...) bench.py must not launch again"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    base = dict(WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    e1 = dict(os.environ); e1.update(base, RANK="1", LOCAL_RANK="1")
    p1 = subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--dry-run"],
                          env=e1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out = _run(["--gpus", "2", "--steps", "2", "--dry-run"], env=dict(base, RANK="0", LOCAL_RANK="0"))
    so, _ = p1.communicate(timeout=120)
    assert p1.returncode == 0
    assert not [ln for ln in so.splitlines() if ln.startswith("{")], "only rank 0 prints"
    assert out["ranks"] == 2 and out["gather_ok"]


def test_usable_cpus_is_affinity_capped():
    sys.path.insert(0, REPO)
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_more_gpus_asked_for_than_visible_is_a_one_line_refusal():
    """`bench.py --gpus N` on a node with fewer GPUs exits non-zero with one line, before any GPU call
    and without launching ranks (this container has none)"""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    msg = [ln for ln in p.stderr.splitlines() if ln.startswith("bench.py:")]
    assert len(msg) == 1 and "--gpus 8" in msg[0] and "nothing was run" in msg[0]
