"""world-size-2 gloo test of the N > 1 path's plumbing (scene ownership, result packing, the one
gather collective, scene-major reordering) -- runs on CPU; the GPU run uses the same code with RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

from neo_planner_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_results(scene, rows, n):
    """stand-in for the optimiser's output of one scene: deterministic in (scene, row)"""
    g = torch.Generator().manual_seed(1000 + scene)
    x = torch.rand(rows, n, generator=g, dtype=torch.float64)
    costs = torch.rand(rows, 4, generator=g, dtype=torch.float64)
    return x, costs


def _worker(rank, world, port, n_scenes, rows, n, q):
    sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = torch.tensor([1.0, 1.0, 1.0, 10000.0], dtype=torch.float64)
    mine = sharding.owned_scenes(n_scenes, rank, world)
    local = torch.cat([sharding.pack_results(*_fake_results(s, rows, n), w) for s in mine])
    allr = sharding.gather_results(local, world)
    # the asynchronous form bench.py uses (several batches in flight): same rows once the work is waited for
    allr2, work = sharding.gather_results(local, world, async_op=True)
    work.wait()
    assert torch.equal(allr, allr2)
    ordered = sharding.scene_major_order(allr, n_scenes, world, rows)
    dist.barrier()
    q.put((rank, mine, ordered.numpy()))
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather():
    world, n_scenes, rows, n = 2, 6, 5, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, rows, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = torch.tensor([1.0, 1.0, 1.0, 10000.0], dtype=torch.float64)
    want = torch.cat([sharding.pack_results(*_fake_results(s, rows, n), w) for s in range(n_scenes)]).numpy()
    owned = {}
    for rank, mine, ordered in got:
        owned[rank] = mine
        assert np.array_equal(ordered, want)            # every rank holds every scene's rows, in scene order
    assert owned[0] == [0, 2, 4] and owned[1] == [1, 3, 5]
    assert sorted(owned[0] + owned[1]) == list(range(n_scenes))
    assert all(sharding.owner_of(s, world) == r for r, ss in owned.items() for s in ss)


def test_pack_results_layout():
    x = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    costs = torch.tensor([[1.0, 2.0, 3.0, 4.0], [0.5, 0.0, 0.0, 1e-3]], dtype=torch.float64)
    w = torch.tensor([1.0, 1.0, 1.0, 10000.0], dtype=torch.float64)
    r = sharding.pack_results(x, costs, w)
    assert r.dtype == torch.float32 and r.shape == (2, 8)
    assert torch.equal(r[:, :3], x.float())
    assert torch.allclose(r[:, 3], torch.tensor([40006.0, 10.5]))
    assert torch.equal(r[:, 4:], costs.float())
    assert sharding.gather_results(r, 1) is r
    assert sharding.gather_results(r, 1, async_op=True) == (r, None) or sharding.gather_results(r, 1, async_op=True)[0] is r


def test_cfg4_scene_split_and_scene_major_round_trip_for_every_world_size():
    """BASELINE.json configs[3]: 256 scenes over the GPUs of a node -- bench.py's `--scenes` default 256 // max(world, 8) per
    rank (32 on 8 GPUs; smaller worlds keep 32 per GPU: weak scaling) -- and for world in {2, 4, 8} the rank-major gathered
    rows go back into scene order (sharding.scene_major_order) for round-robin ownership"""
    import torch
    from neo_planner_amd import sharding
    for world in (2, 4, 8):
        per_gpu = 256 // max(world, 8)
        assert per_gpu == 32
        n_scenes = per_gpu * world
        owned = [sharding.owned_scenes(n_scenes, r, world) for r in range(world)]
        assert sorted(s for o in owned for s in o) == list(range(n_scenes))
        assert all(len(o) == per_gpu for o in owned)
        assert all(sharding.owner_of(s, world) == r for r, o in enumerate(owned) for s in o)
        rows, n = 3, 5
        # what every rank packs: its scenes in increasing order, `rows` result rows each, tagged (scene, row)
        local = [torch.tensor([[s, k, 0, 0, 0] for s in o for k in range(rows)], dtype=torch.float32) for o in owned]
        gathered = torch.cat(local)                       # rank-major, as all_gather_into_tensor leaves it
        back = sharding.scene_major_order(gathered, n_scenes, world, rows)
        assert back.shape == (n_scenes * rows, n)
        assert torch.equal(back[:, 0], torch.arange(n_scenes).repeat_interleave(rows).float())
        assert torch.equal(back[:, 1], torch.arange(rows).repeat(n_scenes).float())


def test_spatial_dispatch_order_is_a_permutation_with_contiguous_runs_per_xcd():
    import numpy as np
    from neo_planner_amd import synth
    from neo_planner_amd.planner import BatchPlanner
    for B in (4096, 1001, 9, 8, 1):
        h, t, _, _ = synth.replan_requests(1, B, 20, D=3, **synth.VOLUME)
        for chunk in (None, 64, 3):
            o = BatchPlanner.spatial_order(h, t, chunk=chunk)
            assert o.dtype == np.int32 and sorted(o.tolist()) == list(range(B))
    h, t, _, _ = synth.replan_requests(1, 4096, 20, D=3, **synth.VOLUME)
    o = BatchPlanner.spatial_order(h, t)
    srt = BatchPlanner.spatial_order(h, t, xcds=1)
    rank = np.empty(4096, dtype=np.int64); rank[srt] = np.arange(4096)
    for k in range(8):                                      # XCD k = workgroups k, k + 8, ...: one contiguous eighth, in order
        r = rank[o[k::8]]
        assert r.min() == k * 512 and r.max() == (k + 1) * 512 - 1 and np.all(np.diff(r) == 1)
