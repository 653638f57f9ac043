"""
Launches with an evaluation budget and resumable runs (include/neo_planner.h
neo_optimize_batch_budget_dev): a trajectory that needs more evaluations than its launch allows is suspended with its
whole optimiser state and finished by later, compacted launches.  The bar: for EVERY trajectory the finished run is the
run of an unbudgeted launch, bit for bit -- x, both sets of cost terms, iteration and evaluation counts, status and the
collision flag -- whatever the budget, in every arithmetic mode.
"""
import ctypes

import numpy as np
import pytest

import neo_planner_amd as npa
from neo_planner_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _scene(layout):
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    dist = synth.esdf_3d(4, n=100, res=0.3, canopy=20)
    g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), 0.3, synth.DOMAIN_ORIGIN, store="f32", layout=layout, ctx=ctx)
    return ctx, g3


@pytest.mark.parametrize("mode,layout", [("f32x", "brick"), ("f64", "brick"), ("f32", "linear")])
def test_budgeted_runs_are_the_unbudgeted_runs_bit_for_bit(mode, layout):
    ctx, g3 = _scene(layout)
    B, M = 700, 21
    head, tail, wp, ts = synth.replan_requests(21, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode)
    x0 = bp.pack_x(wp, ts)
    ref = bp.optimize(g3, x0, head, tail, order=False)
    assert ref["nfev"].max() > 150 and (ref["status"] <= 2).mean() > 0.8
    for budget in (3, 17, 120, 100000):
        got = bp.optimize_budgeted(g3, x0, head, tail, budget)
        for k in ("x", "costs", "costs_last", "nit", "nfev", "status", "collision"):
            assert np.array_equal(got[k], ref[k]), (mode, budget, k, int((got[k] != ref[k]).sum()))
        sizes = got["launch_sizes"]
        assert sizes[0] == B and all(a >= b for a, b in zip(sizes, sizes[1:]))           # compacted re-launches
        if budget >= 100000:
            assert sizes == [B]
        else:
            # a run of nfev evaluations takes ceil(nfev / budget) launches; launch k + 1 holds the runs that need more than k
            need = -(-ref["nfev"].astype(np.int64) // budget)
            assert len(sizes) == int(need.max())
            assert sizes == [int((need > k).sum()) for k in range(len(sizes))]


def test_budget_api_edges():
    import torch
    ctx, g3 = _scene("brick")
    dev = torch.device("cuda", 0)
    B, M, D = 64, 21, 3
    head, tail, wp, ts = synth.replan_requests(5, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
    bp._sync()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0 = t(bp.pack_x(wp, ts)); x = torch.empty_like(x0); h = t(head); tl = t(tail)
    costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
    nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.full_like(nit, -1)
    nbytes = int(ctx.lib.neo_optimize_state_bytes(M, D))
    assert nbytes == 8 * (64 + 5 * 81 + 64 + 2 * 10 * 81)
    state = torch.zeros(B * nbytes, dtype=torch.uint8, device=dev)
    pp = lambda v: ctypes.c_void_p(v.data_ptr())
    call = lambda budget, subset, n_sub, resume: ctx.lib.neo_optimize_batch_budget_dev(
        ctx.h, g3.scene_id, B, M, D, pp(x0), pp(x), pp(h), pp(tl), pp(costs), pp(last), pp(nit), pp(nfev), pp(st), pp(state),
        budget, None if subset is None else pp(subset), n_sub, resume)
    torch.cuda.synchronize()
    assert call(0, None, 0, 0) != 0                                   # a budget below one evaluation
    # a subset launch touches its trajectories only
    sub = torch.tensor([3, 40, 7, B + 5, -1], dtype=torch.int32, device=dev)      # (the last two are outside the arrays: skipped)
    assert call(5, sub, 5, 0) == 0
    ctx.synchronize()
    s = st.cpu().numpy()
    assert set(np.flatnonzero(s != -1).tolist()) == {3, 7, 40}
    assert np.all(s[[3, 7, 40]] == _lib.NEO_TRAJ_SUSPENDED) and np.all(nfev.cpu().numpy()[[3, 7, 40]] == 5)
    # resuming launches continue the suspended ones and leave every other trajectory alone
    assert call(100000, None, 0, 1) == 0
    ctx.synchronize()
    s = st.cpu().numpy()
    assert set(np.flatnonzero(s != -1).tolist()) == {3, 7, 40} and np.all((s[[3, 7, 40]] & 0xff) <= 3)
    ref = bp.optimize(g3, bp.pack_x(wp, ts), head, tail, order=False)
    assert np.array_equal(x.cpu().numpy()[[3, 7, 40]], ref["x"][[3, 7, 40]])
    # n > 128 variables and 2-D maps have no resumable kernel: refused, nothing launched
    m2 = npa.ESDF(ctx=ctx)
    m2.occupancy_map_cb(synth.OccupancyGridMsg(synth.occupancy_2d(1)))
    assert ctx.lib.neo_optimize_batch_budget_dev(ctx.h, m2.scene_id, B, M, D, pp(x0), pp(x), pp(h), pp(tl), pp(costs), pp(last),
                                                 pp(nit), pp(nfev), pp(st), pp(state), 5, None, 0, 0) != 0


@pytest.mark.parametrize("mode", ["f32x", "f64"])
def test_progress_counter_finished_results_are_final_while_the_launch_runs(mode):
    """neo_optimize_progress_counter (round 6): a plain launch counts its trajectories as they finish; what a host copies from
    the result arrays on another stream once the counter says k are complete IS their final result -- x, cost terms, counts,
    status bit for bit what the completed launch leaves -- and the launch computes the same bits with the counter as without"""
    import torch
    ctx, g3 = _scene("brick")
    dev = torch.device("cuda", 0)
    B, M, D = 3000, 21, 3
    head, tail, wp, ts = synth.replan_requests(33, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode)
    bp._sync()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0 = t(bp.pack_x(wp, ts)); hd = t(head); tl = t(tail)
    n = x0.shape[1]

    def bufs():
        return dict(x=torch.empty_like(x0), costs=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                    last=torch.zeros(B, 4, dtype=torch.float64, device=dev), nit=torch.zeros(B, dtype=torch.int32, device=dev),
                    nfev=torch.zeros(B, dtype=torch.int32, device=dev), status=torch.full((B,), -1, dtype=torch.int32, device=dev))
    ref = bufs()
    bp.optimize_dev(g3, ref["x"], hd, tl, ref["costs"], ref["last"], ref["nit"], ref["nfev"], ref["status"], x0=x0)
    ctx.check(ctx.lib.neo_ctx_synchronize(ctx.h))
    got = bufs()
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    h_cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
    snap = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in got.items()}
    torch.cuda.synchronize()
    bp.optimize_dev(g3, got["x"], hd, tl, got["costs"], got["last"], got["nit"], got["nfev"], got["status"], x0=x0, progress=counter)
    seen, snapped_at = [], None
    with torch.cuda.stream(side):
        for _ in range(200000):
            h_cnt.copy_(counter, non_blocking=True)
            side.synchronize()
            k = int(h_cnt[0])
            seen.append(k)
            if snapped_at is None and k >= B // 2:
                # status FIRST: a trajectory marked finished in that copy is final in every array copied after it (one that
                # finishes between two copies shows in the later one only)
                for name in ["status"] + [k_ for k_ in snap if k_ != "status"]:
                    snap[name].copy_(got[name], non_blocking=True)
                    side.synchronize()
                snapped_at = k
            if k >= B:
                break
    ctx.check(ctx.lib.neo_ctx_synchronize(ctx.h))
    ctx.check(ctx.lib.neo_optimize_progress_counter(ctx.h, None))
    assert seen[-1] == B and all(a <= b for a, b in zip(seen, seen[1:]))
    for name in got:                       # the counter changes nothing
        assert torch.equal(got[name], ref[name]), name
    assert snapped_at is not None and B // 2 <= snapped_at <= B
    done = snap["status"].numpy() != -1
    assert int(done.sum()) >= snapped_at               # everything counted had landed (a few more may have by the time of the copy)
    if snapped_at < B:
        assert int(done.sum()) < B or seen[-1] == B    # (the snapshot was taken while the launch was still running, normally)
    for name in got:
        a, b = snap[name].numpy()[done], got[name].cpu().numpy()[done]
        assert np.array_equal(a, b), (name, int((a != b).sum()))


def test_optimize_progressive_yields_every_trajectory_once_with_optimize_s_bits():
    """BatchPlanner.optimize_progressive: the host-array generator on the progress counter"""
    ctx, g3 = _scene("brick")
    B, M = 2500, 21
    head, tail, wp, ts = synth.replan_requests(44, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
    x0 = bp.pack_x(wp, ts)
    ref = bp.optimize(g3, x0, head, tail, order=False)
    got = {k: np.zeros_like(v) for k, v in ref.items()}
    count = np.zeros(B, dtype=np.int64)
    chunks = []
    for idx, res in bp.optimize_progressive(g3, x0, head, tail, shares=(0.5, 0.8, 0.95)):
        count[idx] += 1
        chunks.append(int(idx.size))
        for k in ref:
            got[k][idx] = res[k]
    assert np.all(count == 1) and sum(chunks) == B
    assert chunks[0] >= B // 2 and len(chunks) >= 2            # (the first hand-over holds at least half of the batch)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
