"""Edge cases of the C ABI and the host classes on the GPU: empty and single-element batches, the
smallest and largest shapes, error codes, map updates, foreign map objects, fp16 fields."""
import ctypes
import types

import numpy as np
import pytest

from helpers import golden, load, rel_err

pytestmark = pytest.mark.gpu

import neo_planner_amd as npa
from neo_planner_amd import _lib, synth
from oracle import minco_np as onp


@pytest.fixture(scope="module")
def scene():
    occ = synth.occupancy_2d(6)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    return occ, m, onp.GridESDF(occ, synth.RES, 300, 300, (0.0, -15.0))


def test_empty_batch_is_a_no_op(scene):
    _, m, _ = scene
    bp = npa.BatchPlanner()
    out = bp.cost_grad(m, np.zeros((0, 7)), np.zeros((0, 3, 2)), np.zeros((0, 3, 2)))
    assert out["cost"].shape == (0,)
    res = bp.optimize(m, np.zeros((0, 7)), np.zeros((0, 3, 2)), np.zeros((0, 3, 2)))
    assert res["x"].shape == (0, 7)


def test_error_codes(scene):
    _, m, _ = scene
    c = m.ctx
    x = np.zeros((1, 7)); h = np.zeros((1, 3, 2)); t = np.zeros((1, 3, 2))
    cost = np.zeros(1); c4 = np.zeros((1, 4)); g = np.zeros((1, 7))
    p = _lib.ptr
    # unknown scene
    assert c.lib.neo_cost_grad_batch(c.h, 9999, 1, 3, 2, p(x), p(h), p(t), p(cost), p(c4), p(g), None, None) == 3
    assert b"no ESDF" in c.lib.neo_last_error(c.h)
    # shapes out of range: M > 64, D = 4
    assert c.lib.neo_cost_grad_batch(c.h, m.scene_id, 1, 65, 2, p(x), p(h), p(t), p(cost), p(c4), p(g), None, None) == 1
    assert c.lib.neo_cost_grad_batch(c.h, m.scene_id, 1, 3, 4, p(x), p(h), p(t), p(cost), p(c4), p(g), None, None) == 1
    # null buffers
    assert c.lib.neo_cost_grad_batch(c.h, m.scene_id, 1, 3, 2, None, p(h), p(t), p(cost), p(c4), p(g), None, None) == 1
    # a 3-D field needs D = 3
    vol = np.ones((4, 4, 4), np.float32)
    g3 = npa.ESDF3D(vol, 0.5, (0, 0, 0))
    assert c.lib.neo_cost_grad_batch(c.h, g3.scene_id, 1, 3, 2, p(x), p(h), p(t), p(cost), p(c4), p(g), None, None) == 1
    with pytest.raises(_lib.NeoError):
        npa.BatchPlanner().cost_grad(g3, x, h, t)
    # bad parameters are rejected and leave the old ones in place
    bad = _lib.NeoParams()
    c.lib.neo_params_default(ctypes.byref(bad))
    bad.T_max = bad.T_min
    assert c.lib.neo_params_set(c.h, ctypes.byref(bad)) == 1


def test_single_trajectory_and_largest_shape(scene):
    occ, m, o2 = scene
    rng = np.random.default_rng(4)
    for M, D in ((2, 2), (64, 3), (64, 2), (43, 3)):           # n = 5, 253 (4 slots), 190, 169 (3 -> 4 slots)
        head = np.zeros((1, 3, D)); tail = np.zeros((1, 3, D))
        head[0, 0, :2] = [1.0, 0.5]; tail[0, 0, :2] = [27.0, -1.0]
        if D == 3:
            head[0, 0, 2] = tail[0, 0, 2] = 2.0
        k = np.arange(1, M)[None, :] / M
        wp = (head[0, 0][:, None] + (tail[0, 0] - head[0, 0])[:, None] * k + rng.normal(0, 0.2, (D, M - 1)))[None]
        ts = rng.uniform(0.6, 1.5, (1, M))
        bp = npa.BatchPlanner()
        x = bp.pack_x(wp, ts)
        out = bp.cost_grad(m, x, head, tail, want_coeffs=True)
        pl = onp.OraclePlanner(onp.PlannerParams())
        pl.read_planning_conditions(o2, head[0], tail[0], wp[0], ts[0])
        c = pl.get_cost(x[0]); g = pl.get_grad(x[0])
        assert abs(out["cost"][0] - c) <= 1e-10 * abs(c)
        assert rel_err(out["grad"][0], g) < 1e-9
        assert rel_err(out["coeffs"][0], pl.coeffs) < 1e-10
        res = bp.optimize(m, x, head, tail)
        assert res["status"][0] in (0, 1, 2, 4, 5)        # 4/5: the reference would leave through OverflowError
        if res["status"][0] <= 2:
            assert res["final_cost"][0] <= out["cost"][0] * (1 + 1e-12)


def test_map_update_is_picked_up_and_foreign_maps_are_snapshotted(scene):
    occ, _, _ = scene
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(np.zeros_like(occ)))
    v0 = m.version
    head = np.array([[1.0, 0.0], [0.0, 0.0]]); tail = np.array([[6.0, 0.0], [0.5, 0.0]])
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    pl.plan(m, head, tail)
    free_cost = pl.final_cost
    blocked = np.zeros_like(occ)
    blocked[140:160, 30:40] = 100                                   # a wall across the straight path (y in [-1,1], x in [3,4])
    m.occupancy_map_cb(synth.OccupancyGridMsg(blocked))
    assert m.version == v0 + 1
    pl2 = npa.MinJerkPlanner(npa.PlannerConfig())
    try:
        pl2.plan(m, head, tail)
        assert pl2.final_cost > free_cost                            # has to go around
    except Exception as ex:                                          # or the reference's "No solution" path
        assert "No solution" in str(ex)
    # a map object that only follows the reference's attribute protocol (e.g. the original ESDF class)
    o = onp.GridESDF(blocked, synth.RES, 300, 300, (0.0, -15.0))
    foreign = types.SimpleNamespace(esdf_map=o.esdf_map, esdf_grad_x=o.esdf_grad_x, esdf_grad_y=o.esdf_grad_y,
                                    map_resolution=synth.RES, map_origin=types.SimpleNamespace(x=0.0, y=-15.0),
                                    map_width=300, map_height=300)
    pl3 = npa.MinJerkPlanner(npa.PlannerConfig())
    pl3.read_planning_conditions(foreign, head, tail, *pl3.generate_init_variables(head, tail))
    pl3.tau = pl3.map_T2tau(pl3.ts)
    x = pl3._pack_x()
    ref = onp.OraclePlanner(onp.PlannerParams())
    ref.read_planning_conditions(o, head, tail, pl3.int_wpts, pl3.ts)
    assert abs(pl3.get_cost(x) - ref.get_cost(x)) <= 1e-10 * abs(ref.get_cost(x))
    assert rel_err(pl3.get_grad(x), ref.get_grad(x)) < 1e-9
    assert rel_err(pl3.costs, ref.costs) < 1e-10 or np.allclose(pl3.costs, ref.costs, atol=1e-12)


def test_standalone_cost_terms_like_all_planner_demo(scene):
    """all_planner_demo.py:46-51: get_coeffs, reset_cost, add_sampled_cost on a given (int_wpts, ts)"""
    occ, m, o2 = scene
    head = np.array([[1.0, 0.2], [0.3, 0.0]]); tail = np.array([[6.0, -0.4], [0.6, 0.1]])
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    wp, ts = pl.generate_init_variables(head, tail)
    pl.read_planning_conditions(m, head, tail, wp, ts)
    pl.get_coeffs(wp, ts); pl.reset_cost(); pl.add_sampled_cost(); pl.add_energy_cost(); pl.add_time_cost()
    ref = onp.OraclePlanner(onp.PlannerParams())
    ref.read_planning_conditions(o2, head, tail, wp, ts)
    ref.get_coeffs(wp, ts); ref.reset_cost(); ref.add_sampled_cost(); ref.add_energy_cost(); ref.add_time_cost()
    assert rel_err(pl.coeffs, ref.coeffs) < 1e-11
    assert np.allclose(pl.costs, ref.costs, rtol=1e-10, atol=1e-14)


def test_fp16_field_cost_and_gradient():
    """cfg5 storage: fp16 field, fp32 arithmetic -- against the oracle on the fp16-rounded field"""
    rng = np.random.default_rng(8)
    occ = synth.occupancy_3d(2, n=64, res=30.0 / 64)
    g3 = npa.ESDF3D.from_occupancy(occ, 30.0 / 64, synth.DOMAIN_ORIGIN, store="f16", want_dist=True)
    o3 = onp.Grid3DESDF(g3.dist.astype(np.float16).astype(np.float32), 30.0 / 64, synth.DOMAIN_ORIGIN)
    head, tail, wp, ts = synth.replan_requests(2, 6, 40, D=3)
    bp = npa.BatchPlanner(sample_dtype="f32")
    x = bp.pack_x(wp, ts)
    out = bp.cost_grad(g3, x, head, tail)
    for b in range(6):
        pl = onp.OraclePlanner(onp.PlannerParams())
        pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
        c = pl.get_cost(x[b]); g = pl.get_grad(x[b])
        assert abs(out["cost"][b] - c) <= 5e-5 * abs(c)
        assert rel_err(out["grad"][b], g) < 2e-4


def test_two_contexts_do_not_interfere(scene):
    occ, m, _ = scene
    ctx2 = npa.Context(0)
    m2 = npa.ESDF(ctx2)
    m2.occupancy_map_cb(synth.OccupancyGridMsg(np.zeros_like(occ)))
    head, tail, wp, ts = synth.replan_requests(9, 32, 2, D=2, length_range=(4.0, 6.0))
    bp1 = npa.BatchPlanner(ctx=m.ctx); bp2 = npa.BatchPlanner(ctx=ctx2)
    x = bp1.pack_x(wp, ts)
    a = bp1.optimize(m, x, head, tail); b = bp2.optimize(m2, x, head, tail)
    a2 = bp1.optimize(m, x, head, tail)
    assert np.array_equal(a["x"], a2["x"]) and not np.array_equal(a["x"], b["x"])
    with pytest.raises(ValueError):
        pl = npa.MinJerkPlanner(ctx=ctx2)
        pl.read_planning_conditions(m, head[0], tail[0], wp[0], ts[0])      # map lives on another context
    ctx2.close()


def test_concurrent_callers_on_one_context_are_serialised(scene):
    """rospy calls the planner from timer / action threads: two threads hammering the
    same context must get the answers a single thread gets"""
    import threading
    _, m, _ = scene
    bp = npa.BatchPlanner()
    sets = []
    for k in range(2):
        head, tail, wp, ts = synth.replan_requests(20 + k, 257 + 100 * k, 2 + 18 * k, D=2, length_range=(4.0, 9.0))
        sets.append((bp.pack_x(wp, ts), head, tail))
    want = [bp.cost_grad(m, *a) for a in sets]
    want_opt = [bp.optimize(m, *a) for a in sets]
    errors = []

    def worker(k):
        try:
            for _ in range(15):
                got = bp.cost_grad(m, *sets[k])
                assert np.array_equal(got["grad"], want[k]["grad"]) and np.array_equal(got["cost"], want[k]["cost"])
                opt = bp.optimize(m, *sets[k])
                assert np.array_equal(opt["x"], want_opt[k]["x"])
        except Exception as ex:               # pragma: no cover
            errors.append(repr(ex))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_single_time_evaluators(scene):
    """traj_utils.py:85-179 get_pos / get_vel / get_acc / get_jerk at arbitrary times"""
    _, m, o2 = scene
    head = np.array([[1.0, 0.2], [0.3, 0.0]]); tail = np.array([[6.0, -0.4], [0.6, 0.1]])
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    ref = onp.OraclePlanner(onp.PlannerParams())
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        pl.plan(m, head, tail); ref.plan(o2, head, tail)
    ref.get_coeffs(ref.int_wpts, ref.ts)
    pl.get_coeffs(pl.int_wpts, pl.ts)
    for t in (0.0, 0.37, float(pl.ts[0]), float(sum(pl.ts[:2])) + 0.01, float(sum(pl.ts)), float(sum(pl.ts)) + 3.0):
        assert np.allclose(pl.get_pos(t), ref.get_pos(t), rtol=1e-9, atol=1e-12)
        assert np.allclose(pl.get_vel(t), ref.get_vel(t), rtol=1e-9, atol=1e-12)
        assert np.allclose(pl.get_acc(t), ref.get_acc(t), rtol=1e-8, atol=1e-10)
        assert np.allclose(pl.get_jerk(t), ref.get_jerk(t), rtol=1e-8, atol=1e-9)
    assert pl.get_jer_array().shape == (len(np.arange(0, sum(pl.ts), 0.1)), 2)


def test_two_batches_in_flight_on_two_streams(scene):
    """neo_ctx_set_stream: the same batch issued on two streams of one context (both in flight) gives the
    results of the one-at-a-time run bit for bit"""
    import ctypes
    import torch
    _, m, _ = scene
    ctx = m.ctx
    B, M = 256, 5
    head, tail, wp, ts = synth.replan_requests(5, B, M - 1, D=2, length_range=(6.0, 12.0), jitter=0.3)
    bp = npa.BatchPlanner(ctx=ctx)
    ref = bp.optimize(m, bp.pack_x(wp, ts), head, tail, order=False)
    dev = torch.device("cuda", 0)
    x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    h, tl = torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev)
    torch.cuda.synchronize()
    outs = []
    try:
        for _ in range(2):
            st = torch.cuda.Stream(device=dev)
            ctx.set_stream(st.cuda_stream)
            with torch.cuda.stream(st):
                o = dict(x=x0.clone(), costs=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                         last=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                         nit=torch.zeros(B, dtype=torch.int32, device=dev), nfev=torch.zeros(B, dtype=torch.int32, device=dev),
                         status=torch.zeros(B, dtype=torch.int32, device=dev), st=st)
                bp.optimize_dev(m, o["x"], h, tl, o["costs"], o["last"], o["nit"], o["nfev"], o["status"])
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
    for o in outs:
        assert np.array_equal(o["x"].cpu().numpy(), ref["x"])
        assert np.array_equal(o["nfev"].cpu().numpy(), ref["nfev"])
        assert np.array_equal(o["costs"].cpu().numpy(), ref["costs"])


def test_one_and_two_waves_per_simd_agree_bit_for_bit():
    """NEO_FLAG_ONE_WAVE_PER_SIMD / NEO_FLAG_TWO_WAVES_PER_SIMD select two register allocations of the same
    optimiser kernel: every output must be identical"""
    dist = synth.esdf_3d(2, n=100, res=0.3)
    g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32")
    # M = 31 (n = 121): the L-BFGS pairs leave no room in LDS for the rows of the per-piece fold at eight wavefronts per
    # CU, both variants then fold in registers (neo_launch_opt.hpp)
    for M, B in ((5, 300), (21, 200), (31, 96)):
        head, tail, wp, ts = synth.replan_requests(2, B, M - 1, D=3)
        # ("f64": the parity mode has a two-waves allocation too, for n <= 128 variables)
        for dtype in ("f32", "f64"):
            out = []
            for waves in (1, 2):
                bp = npa.BatchPlanner(sample_dtype=dtype, waves_per_simd=waves)
                out.append(bp.optimize(g3, bp.pack_x(wp, ts), head, tail))
            for k in ("x", "costs", "costs_last", "nit", "nfev", "status"):
                assert np.array_equal(out[0][k], out[1][k]), (M, dtype, k)
            assert out[0]["nfev"].mean() > 10


def test_one_and_two_waves_per_simd_agree_bit_for_bit_on_the_reference_map(scene):
    """the same for the reference's own shape in batches: D = 2 on the nearest-cell map, fp64 and fp32 sampling (from 1024
    trajectories per call on the library takes the two-waves allocation by itself)"""
    _, m, _ = scene
    for M, B in ((3, 200), (5, 128), (21, 64)):
        head, tail, wp, ts = synth.replan_requests(31, B, M - 1, D=2, length_range=(4.0, 9.0), jitter=0.3)
        for dtype in ("f64", "f32"):
            out = []
            for waves in (1, 2):
                bp = npa.BatchPlanner(sample_dtype=dtype, waves_per_simd=waves)
                out.append(bp.optimize(m, bp.pack_x(wp, ts), head, tail))
            for k in ("x", "costs", "costs_last", "nit", "nfev", "status"):
                assert np.array_equal(out[0][k], out[1][k]), (M, dtype, k)
            assert out[0]["nfev"].mean() > 8


def test_two_waves_with_four_slots_store_the_pairs_in_fp32():
    """n > 128 (cfg5's M = 41): the two-waves variant keeps the L-BFGS pairs in fp32 (LDS for eight wavefronts per CU).
    That is a perturbation at the 1e-7 level of an optimiser whose gradient is fp32 already: runs part as any two
    evaluations of this objective do, the batch ends at the same costs with the same effort."""
    dist = synth.esdf_3d(2, n=100, res=0.3)
    g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="yz4")
    M, B = 41, 384
    head, tail, wp, ts = synth.replan_requests(2, B, M - 1, D=3)
    out = []
    for waves in (1, 2):
        bp = npa.BatchPlanner(sample_dtype="f32", waves_per_simd=waves)
        out.append(bp.optimize(g3, bp.pack_x(wp, ts), head, tail))
    a, b = out
    ok = (a["status"] <= 1) & (b["status"] <= 1)
    assert ok.mean() > 0.9
    assert abs(a["nfev"][ok].mean() - b["nfev"][ok].mean()) <= 0.06 * a["nfev"][ok].mean()
    med = np.median(a["final_cost"][ok])
    assert abs(med - np.median(b["final_cost"][ok])) <= 3e-2 * med      # (medians of a few hundred chaotic runs)
    rel = np.abs(a["final_cost"][ok] - b["final_cost"][ok]) / np.abs(a["final_cost"][ok])
    assert np.median(rel) < 2e-2 and (rel < 1e-4).mean() > 0.05
    # first evaluation: no pairs yet, identical
    assert np.array_equal(a["status"] >= 0, b["status"] >= 0)


def test_all_fp32_mode_has_the_statistics_of_the_default_mode():
    """NEO_FLAG_F32_SOLVE ("f32x"): solve, adjoint, optimiser vectors and pairs in fp32.  Per evaluation within 4e-5 of
    the oracle (test_gpu_parity.py); whole runs part from the mixed-precision mode's as any two fp32 evaluations of
    this objective do, the batch ends at the same costs with the same effort, bit-reproducibly."""
    dist = synth.esdf_3d(2, n=100, res=0.3)
    g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="yz4")
    for M, B in ((21, 768), (41, 256), (5, 256)):
        head, tail, wp, ts = synth.replan_requests(5, B, M - 1, D=3)
        ref = npa.BatchPlanner(sample_dtype="f32")
        x0 = ref.pack_x(wp, ts)
        a = ref.optimize(g3, x0, head, tail)
        bx = npa.BatchPlanner(sample_dtype="f32x")
        b, b2 = bx.optimize(g3, x0, head, tail), bx.optimize(g3, x0, head, tail)
        assert np.array_equal(b["x"], b2["x"]) and np.array_equal(b["nfev"], b2["nfev"])
        ok = (a["status"] <= 1) & (b["status"] <= 1)
        assert ok.mean() > 0.9
        assert abs(a["nfev"][ok].mean() - b["nfev"][ok].mean()) <= 0.08 * a["nfev"][ok].mean()
        med = np.median(a["final_cost"][ok])
        assert abs(med - np.median(b["final_cost"][ok])) <= 3e-2 * med      # (medians of a few hundred chaotic runs)
        e0 = bx.cost_grad(g3, x0, head, tail)
        assert np.all(b["final_cost"][ok] <= e0["cost"][ok] * (1 + 1e-6))
    # the lane-group kernel in the same mode (eight small trajectories per wavefront, two wavefronts per SIMD)
    head, tail, wp, ts = synth.replan_requests(6, 2048, 2, D=3, length_range=(4.0, 6.0))
    ga, gb = npa.BatchPlanner(sample_dtype="f32", lane_groups=True), npa.BatchPlanner(sample_dtype="f32x", lane_groups=True)
    x0 = ga.pack_x(wp, ts)
    a, b, b2 = ga.optimize(g3, x0, head, tail), gb.optimize(g3, x0, head, tail), gb.optimize(g3, x0, head, tail)
    assert np.array_equal(b["x"], b2["x"]) and np.array_equal(b["nfev"], b2["nfev"])
    ok = (a["status"] <= 1) & (b["status"] <= 1)
    assert ok.mean() > 0.8                                    # (short requests in a dense scene: more end flagged)
    assert abs(int((a["status"] <= 1).sum()) - int((b["status"] <= 1).sum())) <= 0.05 * len(ok)
    assert abs(a["nfev"][ok].mean() - b["nfev"][ok].mean()) <= 0.08 * a["nfev"][ok].mean()
    med = np.median(a["final_cost"][ok])
    assert abs(med - np.median(b["final_cost"][ok])) <= 3e-2 * med      # (medians of a few hundred chaotic runs)


def test_lane_group_kernel_small_problems():
    """NEO_FLAG_LANE_GROUPS: eight trajectories per wavefront (M = 3, n = 9).  Same algorithm, fp32 sums associated
    differently: runs either follow the default kernel's path (then the results agree to fp32 rounding) or part
    from it the way any two fp32 evaluations of this objective do; the batch as a whole ends at the same costs.
    Results do not depend on which group of which wavefront picks a trajectory up."""
    dist = synth.esdf_3d(2, n=100, res=0.3)
    g_lin = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32")
    g_yz4 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="yz4")
    for M, B, g3 in ((3, 1001, g_lin), (3, 1001, g_yz4), (4, 130, g_lin), (2, 64, g_yz4), (3, 1, g_lin), (6, 200, g_yz4),
                     (8, 65, g_lin)):
        head, tail, wp, ts = synth.replan_requests(4, B, M - 1, D=3, length_range=(4.0, 6.0))
        ref = npa.BatchPlanner(sample_dtype="f32")
        x0 = ref.pack_x(wp, ts)
        a = ref.optimize(g3, x0, head, tail, order=False)
        grp = npa.BatchPlanner(sample_dtype="f32", lane_groups=True)
        b = grp.optimize(g3, x0, head, tail, order=False)
        b2 = grp.optimize(g3, x0, head, tail, order=False)
        for k in ("x", "costs", "costs_last", "nit", "nfev", "status"):
            assert np.array_equal(b[k], b2[k]), (M, k)              # whichever group took which trajectory
        ok = (a["status"] <= 1) & (b["status"] <= 1)
        same = ok & (a["nit"] == b["nit"]) & (a["nfev"] == b["nfev"])
        if B >= 1000:
            assert same.mean() >= 0.3, (M, same.mean())
        if B >= 64:
            assert abs(np.median(a["final_cost"][ok]) - np.median(b["final_cost"][ok])) <= 2e-2 * np.median(a["final_cost"][ok])
            assert abs(int((a["status"] <= 1).sum()) - int((b["status"] <= 1).sum())) <= max(6, 0.05 * B)     # (statuses 2 and 4 come and go with the path a run takes: DESIGN.md section 3)
        if same.any():
            # (same path, fp32-level differences in every evaluation: a few runs end a little apart along flat
            #  directions of the objective)
            assert np.percentile(np.abs(a["x"][same] - b["x"][same]).max(axis=1), 90) < 1e-2
            rel = np.abs(a["final_cost"][same] - b["final_cost"][same]) / np.abs(a["final_cost"][same])
            assert np.median(rel) < 1e-5
    # shapes the group kernel does not cover fall back to the default kernel: bit-identical then
    head, tail, wp, ts = synth.replan_requests(4, 40, 20, D=3)
    ref = npa.BatchPlanner(sample_dtype="f32")
    x0 = ref.pack_x(wp, ts)
    a = ref.optimize(g_lin, x0, head, tail)
    b = npa.BatchPlanner(sample_dtype="f32", lane_groups=True).optimize(g_lin, x0, head, tail)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["nfev"], b["nfev"])
    # the two layouts hold the same numbers: the same sums in the same order
    la = npa.BatchPlanner(sample_dtype="f32", lane_groups=True)
    head, tail, wp, ts = synth.replan_requests(4, 256, 2, D=3, length_range=(4.0, 6.0))
    x0 = la.pack_x(wp, ts)
    a, b = la.optimize(g_lin, x0, head, tail, order=False), la.optimize(g_yz4, x0, head, tail, order=False)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["nfev"], b["nfev"])
    g_brk = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="brick")
    c = la.optimize(g_brk, x0, head, tail, order=False)
    assert np.array_equal(a["x"], c["x"]) and np.array_equal(a["nfev"], c["nfev"])


def test_lane_groups_with_dispatch_order_and_batches_in_flight():
    """the ticket counters of concurrent lane-group launches are separate, and a dispatch order only changes who
    starts first: results identical to the plain one-at-a-time run"""
    import ctypes
    import torch
    dist = synth.esdf_3d(2, n=100, res=0.3)
    dev = torch.device("cuda", 0)
    ctx = npa.Context(0)
    g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
    B, M = 3000, 3
    head, tail, wp, ts = synth.replan_requests(9, B, M - 1, D=3, length_range=(4.0, 6.0))
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32", lane_groups=True)
    ref = bp.optimize(g3, bp.pack_x(wp, ts), head, tail, order=False)
    x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    h, tl = torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev)
    order = torch.from_numpy(np.random.default_rng(0).permutation(B).astype(np.int32)).to(dev)
    ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(order.data_ptr()), B))
    torch.cuda.synchronize()
    outs = []
    try:
        for _ in range(3):
            st = torch.cuda.Stream(device=dev)
            ctx.set_stream(st.cuda_stream)
            with torch.cuda.stream(st):
                o = dict(x=x0.clone(), costs=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                         last=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                         nit=torch.zeros(B, dtype=torch.int32, device=dev), nfev=torch.zeros(B, dtype=torch.int32, device=dev),
                         status=torch.zeros(B, dtype=torch.int32, device=dev), st=st)
                bp.optimize_dev(g3, o["x"], h, tl, o["costs"], o["last"], o["nit"], o["nfev"], o["status"])
            outs.append(o)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
        ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, None, 0))
    for o in outs:
        assert np.array_equal(o["x"].cpu().numpy(), ref["x"])
        assert np.array_equal(o["nfev"].cpu().numpy(), ref["nfev"])
        assert np.array_equal(o["costs"].cpu().numpy(), ref["costs"])
        assert np.array_equal(o["status"].cpu().numpy() & 0xff, ref["status"])


def test_map_update_while_batches_are_in_flight():
    """A map update on one thread/stream while `_dev` batches of the same scene are in flight on other streams
    (the record buffer of a same-size update is rewritten in place): the update waits for the device, so
    every batch launched BEFORE it sees the old map (results = the old map's, bit for bit) and a batch launched AFTER
    it the new one; nothing reads a half-written map."""
    import torch
    occ_a, occ_b = synth.occupancy_2d(21), synth.occupancy_2d(22)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ_a))
    ctx = m.ctx
    B, M = 2048, 5
    head, tail, wp, ts = synth.replan_requests(9, B, M - 1, D=2, length_range=(8.0, 16.0), jitter=0.3)
    bp = npa.BatchPlanner(ctx=ctx)
    ref_a = bp.optimize(m, bp.pack_x(wp, ts), head, tail, order=False)
    dev = torch.device("cuda", 0)
    x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    h, tl = torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev)

    def issue(st):
        ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            o = dict(x=x0.clone(), costs=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                     last=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                     nit=torch.zeros(B, dtype=torch.int32, device=dev), nfev=torch.zeros(B, dtype=torch.int32, device=dev),
                     status=torch.zeros(B, dtype=torch.int32, device=dev))
            bp.optimize_dev(m, o["x"], h, tl, o["costs"], o["last"], o["nit"], o["nfev"], o["status"])
        return o
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    try:
        before = [issue(st) for st in streams for _ in range(2)]        # six batches in flight on three streams
        ctx.set_stream(None)
        m.occupancy_map_cb(synth.OccupancyGridMsg(occ_b))               # same size: rewrites the record buffer in place
        after = issue(streams[0])
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
    for o in before:
        assert np.array_equal(o["x"].cpu().numpy(), ref_a["x"])
        assert np.array_equal(o["nfev"].cpu().numpy(), ref_a["nfev"])
    ref_b = bp.optimize(m, bp.pack_x(wp, ts), head, tail, order=False)
    assert np.array_equal(after["x"].cpu().numpy(), ref_b["x"])
    assert not np.array_equal(ref_a["x"], ref_b["x"])


def test_multi_scene_calls_reject_mixed_maps_and_bad_slots():
    """per-trajectory scenes must all be of one kind / element type / layout (one kernel instantiation serves
    the call); the host path checks the ids, the device path checks the context's maps and bounds the slots"""
    import torch
    ctx = _lib.Context(0)
    occ3 = synth.occupancy_3d(0, n=48, res=30.0 / 48)
    a = npa.ESDF3D.from_occupancy(occ3, 30.0 / 48, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
    b = npa.ESDF3D.from_occupancy(occ3, 30.0 / 48, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
    B, M = 64, 4
    head, tail, wp, ts = synth.replan_requests(2, B, M - 1, D=3, length_range=(8.0, 14.0))
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32")
    ids = np.where(np.arange(B) % 2 == 0, a.scene_id, b.scene_id).astype(np.int32)
    ok = bp.optimize(a, bp.pack_x(wp, ts), head, tail, scene_ids=ids)
    one = bp.optimize(a, bp.pack_x(wp, ts), head, tail)
    assert np.array_equal(ok["x"], one["x"])                       # both scenes hold the same field
    # a third scene with another element type makes multi-scene calls ambiguous -> rejected, not mis-dispatched
    c16 = npa.ESDF3D.from_occupancy(occ3, 30.0 / 48, synth.DOMAIN_ORIGIN, store="f16", ctx=ctx)
    with pytest.raises(_lib.NeoError):
        bp.optimize(a, bp.pack_x(wp, ts), head, tail, scene_ids=ids)
    ctx.check(ctx.lib.neo_esdf_drop(ctx.h, c16.scene_id))
    # mixing a 2-D map into the ids
    m2 = npa.ESDF(ctx=ctx)
    m2.occupancy_map_cb(synth.OccupancyGridMsg(synth.occupancy_2d(1)))
    mixed = ids.copy(); mixed[3] = m2.scene_id
    with pytest.raises(_lib.NeoError):
        bp.optimize(a, bp.pack_x(wp, ts), head, tail, scene_ids=mixed)
    # a slot outside the table: that trajectory is flagged NEO_TRAJ_BAD_SCENE and left untouched, the others run
    dev = torch.device("cuda", 0)
    bp._sync()
    slots = torch.full((B,), ctx.lib.neo_scene_slot(ctx.h, a.scene_id), dtype=torch.int32, device=dev)
    slots[5] = 77
    x = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    x_in = x.clone()
    costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
    nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
    bp.optimize_dev(a, x, torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev), costs, last, nit, nfev, st, slots=slots)
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert (st[5] & 0xff) == _lib.NEO_TRAJ_BAD_SCENE and torch.equal(x[5], x_in[5])
    assert ((np.delete(st, 5) & 0xff) <= 5).all()              # ordinary terminations (4: a run that left through exp overflow)
    assert np.array_equal(np.delete(x.cpu().numpy(), 5, axis=0), np.delete(one["x"], 5, axis=0))


def test_batched_plan_with_retries_like_warm_start_plan():
    """BatchPlanner.plan: the reference's retry loop (expert_planner.py:186-203) for a batch -- failed requests (collision
    cost too large / overflow) are re-seeded with N(0, 0.5) jitter and launched again, the others keep their first
    answer bit for bit; requests through a pillar need retries, free ones do not."""
    occ = synth.occupancy_2d(3, count=40)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    boxes = synth.forest_boxes(3, count=40)
    heads, tails = [], []
    for (cx, cy, *_rest) in boxes[:24]:
        heads.append([[cx - 2.5, cy], [0.5, 0.0]]); tails.append([[cx + 2.5, cy], [0.5, 0.0]])       # straight through a pillar
    for k in range(24):
        heads.append([[0.5, -14.0 + 0.1 * k], [0.0, 0.0]]); tails.append([[4.5, -14.0 + 0.1 * k], [0.6, 0.0]])   # free corridor
    head = np.zeros((48, 3, 2)); tail = np.zeros((48, 3, 2))
    head[:, :2] = np.array(heads); tail[:, :2] = np.array(tails)
    free = np.array([not (m.has_collision(h[0]) or m.has_collision(t[0])) for h, t in zip(head, tail)])
    bp = npa.BatchPlanner()
    out = bp.plan(m, head, tail, max_attempts=5, rng=np.random.default_rng(7))
    first = bp.optimize(m, bp.pack_x(*bp.init_guess(head, tail, 2)), head, tail)
    assert out["attempts"].min() == 1 and out["attempts"].max() <= 5
    once = out["attempts"] == 1
    assert np.array_equal(out["x"][once], first["x"][once]) and once[24:].all()
    assert (out["attempts"][:24] > 1).sum() >= 4                     # some pillar requests needed the re-seeded attempts
    retried = out["attempts"] > 1
    assert out["solved"][retried & free].mean() >= 0.5               # ... and most of those found a way round
    assert np.all(out["final_cost"][out["solved"]] < 1e4)
    # a request's re-seeded attempts do not depend on its batch neighbours: the pillar requests alone, same
    # seed -> the same answers, although other requests fail around them in the full batch
    a = bp.plan(m, head, tail, max_attempts=5, seed=11)
    sub = np.r_[0:24:2]
    b = bp.plan(m, head[sub], tail[sub], max_attempts=5, seed=11)
    # (request i of the sub-batch has index i there: its stream is keyed by the index, so compare equal indices)
    c = bp.plan(m, head[:12], tail[:12], max_attempts=5, seed=11)
    assert np.array_equal(a["x"][:12], c["x"][:12]) and np.array_equal(a["attempts"][:12], c["attempts"][:12])
    assert b["x"].shape[0] == 12
    with pytest.raises(ValueError):
        bp.plan(m, head, tail, int_wpts=np.zeros((48, 2, 2)))         # ts missing
    # the initial guess is the reference's (generate_init_variables, fixed mode, :82-101)
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    w, t = pl.generate_init_variables(head[0, :2], tail[0, :2])
    wb, tb = bp.init_guess(head[:1], tail[:1], 2)
    assert np.allclose(wb[0], w, rtol=1e-14, atol=0) and np.array_equal(tb[0], t)


def test_dispatch_order_changes_nothing_but_the_order_of_execution():
    """neo_optimize_dispatch_order (effort order of the optimiser, XCD-aware spatial order of the ESDF-lookup kernel,
    BatchPlanner.spatial_order): results stay in the caller's order and are bit-identical under any permutation, for the
    stand-alone sampled-terms kernel and for whole optimiser runs; a permutation of another batch size is ignored"""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    dist = synth.esdf_3d(2, n=100, res=0.3, canopy=20)
    ctx = _lib.Context(0)
    g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
    B, M, D = 1000, 21, 3
    head, tail, wp, ts = synth.replan_requests(7, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
    x0 = bp.pack_x(wp, ts)
    e = bp.cost_grad(g3, x0, head, tail, want_coeffs=True)
    ref_s = bp.sampled_terms(g3, e["coeffs"], ts)
    ref_o = bp.optimize(g3, x0, head, tail, order=False)
    rng = np.random.default_rng(3)
    for perm in (npa.BatchPlanner.spatial_order(head, tail), npa.BatchPlanner.spatial_order(head, tail, chunk=16),
                 rng.permutation(B).astype(np.int32)):
        assert sorted(perm.tolist()) == list(range(B))
        ctx.check(ctx.lib.neo_optimize_dispatch_order_host(ctx.h, _lib.ptr(np.ascontiguousarray(perm, dtype=np.int32)), B))
        got = bp.sampled_terms(g3, e["coeffs"], ts, order=perm)
        for k in ("costs2", "grad_C", "grad_T"):
            assert np.array_equal(got[k], ref_s[k]), k
        # (BatchPlanner.optimize installs its own order; drive the ABI directly to keep this one)
        x = x0.copy()
        costs = np.zeros((B, 4)); last = np.zeros((B, 4)); nit = np.zeros(B, np.int32); nfev = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
        ctx.check(ctx.lib.neo_optimize_batch(ctx.h, g3.scene_id, None, B, M, D, _lib.ptr(x), _lib.ptr(head), _lib.ptr(tail), _lib.ptr(costs),
                                             _lib.ptr(last), _lib.ptr(nit), _lib.ptr(nfev), _lib.ptr(st)))
        assert np.array_equal(x, ref_o["x"]) and np.array_equal(nfev, ref_o["nfev"]) and np.array_equal(costs, ref_o["costs"])
    # a permutation given for another batch size does not apply; the ESDF-lookup kernel's order is the context's own copy
    # (neo_sampled_terms_dispatch_order): a device array handed over may be freed right after the call
    pd = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int32)).to(dev)
    ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, ctypes.c_void_p(pd.data_ptr()), 1, B))
    pd.fill_(-1 << 20)       # (what a stale pointer would make the kernel index with)
    del pd
    ctx.synchronize()
    coeffs_d = torch.from_numpy(e["coeffs"]).to(dev); ts_d = torch.from_numpy(np.ascontiguousarray(ts)).to(dev)
    c2 = torch.zeros(B, 2, dtype=torch.float64, device=dev); gC = torch.zeros(B, 6 * M, D, dtype=torch.float64, device=dev)
    gT = torch.zeros(B, M, dtype=torch.float64, device=dev)
    pp = lambda t_: ctypes.c_void_p(t_.data_ptr())
    ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(coeffs_d), pp(ts_d), pp(c2), pp(gC), pp(gT)))
    ctx.synchronize()
    assert np.array_equal(gC.cpu().numpy(), ref_s["grad_C"]) and np.array_equal(c2.cpu().numpy(), ref_s["costs2"])
    ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, _lib.ptr(np.ascontiguousarray(perm, dtype=np.int32)), 0, B))
    c2h = np.zeros((500, 2)); gCh = np.zeros((500, 6 * M, D)); gTh = np.zeros((500, M))
    ctx.check(ctx.lib.neo_sampled_terms_batch(ctx.h, g3.scene_id, 500, M, D, _lib.ptr(np.ascontiguousarray(e["coeffs"][:500])),
                                              _lib.ptr(np.ascontiguousarray(ts[:500])), _lib.ptr(c2h), _lib.ptr(gCh), _lib.ptr(gTh)))
    assert np.array_equal(gCh, ref_s["grad_C"][:500])
    ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, None, 0, 0))
    ctx.check(ctx.lib.neo_optimize_dispatch_order_host(ctx.h, None, 0))


def test_optimiser_and_evaluation_kernels_compute_the_same_bits_in_the_all_fp32_mode():
    """The all-fp32 optimiser kernel keeps the cyclic reduction's multipliers in dynamic LDS and folds the sampled partials
    through per-piece accumulators; the evaluation kernel keeps them in static LDS and folds through rows.  Both must be
    the same arithmetic: every point a run evaluates (neo_optimize_trace_xg) gives, re-evaluated by neo_cost_grad_batch,
    the gradient the run saw, bit for bit.  With the multiplier reuse switched off (flags bit 4096: the adjoint reduces
    K^T itself) the gradients agree to fp32 round-off and the batch statistics are the same."""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    dist = synth.esdf_3d(5, n=120, res=0.25, canopy=30)
    g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), 0.25, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
    B, M, D, CAP = 64, 21, 3, 600
    n = D * (M - 1) + M
    head, tail, wp, ts = synth.replan_requests(11, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
    bp._sync()
    x0 = bp.pack_x(wp, ts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_x0, d_x, d_h, d_t = t(x0), torch.empty(B, n, dtype=torch.float64, device=dev), t(head), t(tail)
    costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
    nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
    xg = torch.zeros(B, CAP, 2, n, dtype=torch.float64, device=dev)
    ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, ctypes.c_void_p(xg.data_ptr()), CAP))
    bp.optimize_dev(g3, d_x, d_h, d_t, costs, last, nit, nfev, st, x0=d_x0)
    ctx.synchronize()
    ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, None, 0))
    nf = nfev.cpu().numpy(); xgh = xg.cpu().numpy()
    checked = 0
    for b in range(0, B, 4):
        E = min(int(nf[b]), CAP)
        pts = xgh[b, :E, 0]
        e = bp.cost_grad(g3, pts, np.repeat(head[b:b + 1], E, axis=0), np.repeat(tail[b:b + 1], E, axis=0))
        ok = e["status"] == 0
        assert np.array_equal(e["grad"][ok], xgh[b, :E, 1][ok]), b
        checked += int(ok.sum())
    assert checked > 1000
    # the same batch without the reuse
    bq = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
    bq.flags |= 4096
    r_on = bp.optimize(g3, x0, head, tail)
    r_off = bq.optimize(g3, x0, head, tail)
    e_on = bp.cost_grad(g3, x0, head, tail)       # (the evaluation kernel reuses whatever the flags say nothing about: compare the optimisers)
    assert abs(r_on["nfev"].mean() - r_off["nfev"].mean()) <= 0.1 * r_off["nfev"].mean()
    both = (r_on["status"] <= 1) & (r_off["status"] <= 1)
    assert both.mean() > 0.7
    assert abs(np.median(r_on["final_cost"][both]) - np.median(r_off["final_cost"][both])) <= 2e-2 * np.median(r_off["final_cost"][both])
    assert np.all(np.isfinite(e_on["grad"]))


def test_all_fp32_multiplier_reuse_for_every_piece_count_and_both_maps():
    """The adjoint from the forward reduction's multipliers, the LDS exchange table and the accumulator fold, for every
    piece count of the lane = (piece, dimension) layout (M = 2 .. 21 at D = 3, up to 32 on the 2-D map with D = 2) and
    beyond it (lane = piece: exchange table only): against the same runs with the reuse switched off (flags bit 4096) --
    per evaluation the two gradients agree to fp32 round-off, whole batches end with the same statistics, nothing is
    ever non-finite -- after a kernel that leaves NaN patterns in LDS (unwritten-slot reads would pick them up)."""
    import torch
    rng = np.random.default_rng(5)
    ctx = _lib.Context(0)
    dev = torch.device("cuda", 0)
    dist = synth.esdf_3d(4, n=100, res=0.3, canopy=20)
    g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
    occ = synth.occupancy_2d(4, count=30)
    m2 = npa.ESDF(ctx=ctx)
    m2.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    # poison LDS: a kernel whose staging holds NaN bit patterns at exit (the sampled-terms kernel fed NaN coefficients)
    nanc = np.full((64, 6 * 21, 3), np.nan)
    npa.BatchPlanner(ctx=ctx, sample_dtype="f32").sampled_terms(g3, nanc, np.full((64, 21), 2.0))
    for D, scene, Ms in ((3, g3, (2, 3, 4, 5, 8, 9, 16, 17, 20, 21, 22, 31)), (2, m2, (2, 3, 7, 16, 21, 31, 32, 33))):
        for M in Ms:
            B = 96
            head, tail, wp, ts = synth.replan_requests(100 + M, B, M - 1, D=D, length_range=(6.0, 20.0))
            on = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
            off = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
            off.flags |= 4096
            x0 = on.pack_x(wp, ts)
            e_on, e_off = on.cost_grad(scene, x0, head, tail), off.cost_grad(scene, x0, head, tail)
            assert np.all(np.isfinite(e_on["grad"])) and np.all(np.isfinite(e_on["cost"])), (D, M)
            scale = np.abs(e_off["grad"]).max(axis=1)
            assert np.all(np.abs(e_on["grad"] - e_off["grad"]).max(axis=1) <= 2e-4 * scale + 1e-6), (D, M)
            r_on, r_off = on.optimize(scene, x0, head, tail), off.optimize(scene, x0, head, tail)
            assert np.all(np.isfinite(r_on["x"])) and set(np.unique(r_on["status"])) <= {0, 1, 2, 4}, (D, M, np.unique(r_on["status"]))
            assert abs(int((r_on["status"] <= 1).sum()) - int((r_off["status"] <= 1).sum())) <= 0.1 * B + 3, (D, M)
            assert abs(r_on["nfev"].mean() - r_off["nfev"].mean()) <= 0.25 * r_off["nfev"].mean() + 3, (D, M)
            both = (r_on["status"] <= 1) & (r_off["status"] <= 1)
            if both.sum() >= 20:
                # (a hundred chaotic runs whose costs span three orders of magnitude in two clusters -- collision-free or not:
                #  the median jumps between the clusters with a handful of runs; the geometric mean moves with their share)
                lg_on, lg_off = np.log10(r_on["final_cost"][both]), np.log10(r_off["final_cost"][both])
                assert abs(lg_on.mean() - lg_off.mean()) <= 0.4, (D, M, lg_on.mean(), lg_off.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f64", "f32", "f32x"])
def test_a_non_finite_start_point_ends_its_own_run_only(mode):
    """a NaN in one request's start point: that run ends at its first evaluation with NEO_TRAJ_NONFINITE (the objective is
    NaN: L-BFGS-B has nothing to line-search on; NEO_TRAJ_NUMERIC_RANGE in the all-fp32 mode), x left as given; every other run of the batch is the run it is without
    that request, bit for bit -- in all three arithmetic modes (the all-fp32 one carries f, the step and the line-search
    state in fp32 since round 5)."""
    grid = 96
    res = 30.0 / grid
    occ = synth.occupancy_3d(2, n=grid, res=res, canopy=10)
    g3 = npa.ESDF3D.from_occupancy(occ, res, synth.DOMAIN_ORIGIN, store="f32", layout="brick")
    M, B = 21, 96
    h, t, w, ts = synth.replan_requests(7, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(sample_dtype=mode)
    x0 = bp.pack_x(w, ts)
    good = bp.optimize(g3, x0, h, t)
    bad = x0.copy()
    bad[17, 5] = np.nan
    out = bp.optimize(g3, bad, h, t)
    # NEO_TRAJ_NONFINITE; the all-fp32 evaluation reports NEO_TRAJ_NUMERIC_RANGE: there a jerk outside the fp32 range shows
    # as inf or NaN and either raises the range flag (neo_device.hpp minco_backward), before the objective is looked at
    assert out["status"][17] == (4 if mode == "f32x" else 5)
    assert out["nfev"][17] == 1 and out["nit"][17] == 0
    keep = np.arange(B) != 17
    assert np.array_equal(out["x"][keep], good["x"][keep])
    assert np.array_equal(out["nfev"][keep], good["nfev"][keep]) and np.array_equal(out["status"][keep], good["status"][keep])
    assert np.array_equal(out["costs"][keep], good["costs"][keep])


def test_effort_order_on_the_device_is_the_host_order():
    """neo_effort_order_dev (round 6): the expected-effort dispatch order from resident buffers -- a permutation, keys
    non-increasing along it with ties in index order, and BatchPlanner.expected_effort_order's permutation wherever the host's
    keys are not within rounding of a neighbour's; B not a multiple of the tile, equal keys, a NaN start point"""
    import torch
    dev = torch.device("cuda", 0)
    for B, M, D in ((4096, 21, 3), (1000, 21, 3), (257, 3, 2), (3, 41, 3)):
        head, tail, wp, ts = synth.replan_requests(5, B, M - 1, D=D, **(synth.VOLUME if D == 3 else {}))
        ts[B // 2:B // 2 + 2] = ts[0]; head[B // 2:B // 2 + 2] = head[0]; tail[B // 2:B // 2 + 2] = tail[0]      # equal keys
        bp = npa.BatchPlanner(sample_dtype="f32")
        x0 = bp.pack_x(wp, ts)
        if B > 100:
            x0[7, -1] = np.nan
        order, keys = bp.expected_effort_order_dev(torch.from_numpy(x0).to(dev), torch.from_numpy(head).to(dev),
                                                   torch.from_numpy(tail).to(dev))
        torch.cuda.synchronize()
        order = order.cpu().numpy(); keys = keys.cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(B))
        ko = keys[order]
        assert np.all(ko[:-1] >= ko[1:])
        same = ko[:-1] == ko[1:]
        assert np.all(order[:-1][same] < order[1:][same])
        dist = np.linalg.norm(tail[:, 0] - head[:, 0], axis=1)
        T = bp.unpack_x(x0, M, D)[1]
        slack = T.sum(axis=1) * bp.cfg.v_max / np.maximum(dist, 1e-9)
        ok = np.isfinite(slack)
        assert np.allclose(keys[ok], slack[ok], rtol=1e-13, atol=0)
        if B > 100:
            assert keys[7] == 0.0 and order[-1] == 7 or keys[order[-1]] == 0.0
