"""
Parity tests proper (run on the MI355X box with -m gpu).  Everything goes through the C ABI
(ctypes -> libneo_planner_hip.so); the oracle (oracle/minco_np.py, pinned to the reference by
tests/test_oracle_golden.py) and the committed reference fixtures are the checkers.

Tolerances
  fp64 sampling mode: per-evaluation cost / gradient / coefficients 1e-10 relative (round-off of
      two different but exact solution methods, cond <= 4e5); optimiser results 1e-4 relative
      (BASELINE.json) on every recorded reference run bar the two named in KNOWN_PARTED, in practice 1e-12
      whenever the run is not decided by round-off.
  fp32 sampling mode: per-evaluation 2e-5 relative (all-fp32 mode "f32x": 4e-5); optimiser: final cost statistics only.
  ESDF construction and lookups: bit-exact.
"""
import contextlib
import io

import numpy as np
import pytest

from helpers import golden, load, rel_err

pytestmark = pytest.mark.gpu

import neo_planner_amd as npa
from neo_planner_amd import _lib, synth
from oracle import minco_np as onp


def _gpu_map(d):
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(d["occ"], float(d["res"]), d["origin"]))
    return m


def _oracle_map(d):
    occ = d["occ"]
    return onp.GridESDF(occ, float(d["res"]), occ.shape[1], occ.shape[0], d["origin"])


# ----------------------------------------------------------------------------- maps
@pytest.mark.parametrize("path", golden("g2_esdf_*.npz"))
def test_esdf_build_and_lookup_bit_exact(path):
    d = load(path)
    m = _gpu_map(d)
    assert np.array_equal(m.esdf_map, d["esdf_map"])
    assert np.array_equal(m.esdf_grad_x, d["esdf_grad_x"])
    assert np.array_equal(m.esdf_grad_y, d["esdf_grad_y"])
    dis, grd = m.query(d["probe_pts"])
    assert np.array_equal(dis, d["probe_dis"])
    assert np.array_equal(grd, d["probe_grad"])
    for p, want_d, want_g, want_c in list(zip(d["probe_pts"], d["probe_dis"], d["probe_grad"], d["probe_collision"]))[::17]:
        assert float(m.get_edt_dis(p)) == want_d
        assert [float(v) for v in m.get_edt_grad(p)] == list(want_g)
        assert bool(m.has_collision(p)) == bool(want_c)


@pytest.mark.parametrize("shape,seed", [((300, 300), 0), ((2, 7), 1), ((9, 2), 2), ((64, 257), 3), ((5, 5), 4),
                                        ((600, 40), 5), ((3, 520), 6), ((37, 700), 7)])
def test_esdf_build_matches_scipy_on_odd_shapes(shape, seed):
    """maps up to 512 x 512 take the exhaustive per-cell kernels, larger ones the lower-envelope sweeps
    (neo_abi.hip: edt2_*_bf / edt_columns + edt_rows): both paths, both with and without obstacles"""
    rng = np.random.default_rng(seed)
    occ = np.where(rng.random(shape) < 0.03, 100, 0).astype(np.int8)
    if seed in (4, 6):
        occ[:] = 0                       # no obstacle at all: scipy's virtual background corner
    if seed == 0:
        occ = synth.occupancy_2d(7, unknown_frac=0.05)
    o = onp.GridESDF(occ, 0.1, occ.shape[1], occ.shape[0], (0.5, -2.0))
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ, 0.1, (0.5, -2.0)))
    assert np.array_equal(m.esdf_map, o.esdf_map)
    assert np.array_equal(m.esdf_grad_x, o.esdf_grad_x)      # (np.gradient needs >= 2 cells per axis)
    assert np.array_equal(m.esdf_grad_y, o.esdf_grad_y)


@pytest.mark.parametrize("shape,seed", [((48, 48, 48), 0), ((20, 33, 47), 1), ((5, 64, 9), 2),
                                        # long lines: the narrower LDS tiles of the y / z passes (16, 8 and 2 columns)
                                        ((6, 600, 70), 3), ((700, 6, 40), 4), ((4, 3000, 10), 5), ((2, 2, 2), 6),
                                        # z lines whose tile would pass 64 KB only together with the kernel's static LDS
                                        ((680, 5, 20), 7), ((1360, 3, 9), 8)])
def test_esdf_build_3d_is_the_exact_edt(shape, seed):
    """device-side 3-D EDT against scipy.ndimage.distance_transform_edt: equal after the same fp32 rounding"""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    occ = (rng.random(shape) < 0.01).astype(np.uint8)
    occ[0] = 1                                            # ground slab, as every synthetic scene has
    want = (ndimage.distance_transform_edt(1 - occ) * 0.1).astype(np.float32)
    for layout in ("linear", "yz4", "cell8", "brick"):
        g3 = npa.ESDF3D.from_occupancy(occ, 0.1, (0.0, -1.0, 0.0), layout=layout, want_dist=True)
        assert np.array_equal(g3.dist, want)
        pts = rng.uniform([0, -1, 0], [shape[2] * 0.1, -1 + shape[1] * 0.1, shape[0] * 0.1], (500, 3))
        o3 = onp.Grid3DESDF(want, 0.1, (0.0, -1.0, 0.0))
        dis, _ = g3.query(pts)
        assert np.max(np.abs(dis - np.array([o3.lookup(p)[0] for p in pts]))) < 1e-12


def test_esdf_build_3d_packed_key_and_general_line_passes_agree():
    """round 4: the y / z passes run on packed (cost, minimiser) keys where the volume's squared diagonal leaves room in 31
    bits, and in the general form elsewhere (neo_esdf_build_config forces it); the x pass has a form with 8 / 16 voxels per
    lane for rows of 4-byte aligned length.  Every form against SciPy, on the shapes that take the fast forms by default."""
    from scipy import ndimage
    rng = np.random.default_rng(21)
    for shape, dens in (((48, 48, 48), 0.01), ((33, 300, 64), 0.002), ((300, 20, 40), 0.002), ((9, 40, 600), 0.001),
                        ((12, 40, 1028), 0.001), ((21, 33, 47), 0.02),
                        # x rows at the edges of the 8 / 16 voxels-a-lane forms: one load a lane, all 64 lanes, one past
                        ((5, 6, 4), 0.05), ((6, 9, 512), 0.002), ((6, 9, 516), 0.002), ((5, 7, 1024), 0.001)):
        occ = (rng.random(shape) < dens).astype(np.uint8)
        occ[0] = 1
        occ[:, shape[1] // 2:, :] &= (rng.random((shape[0], shape[1] - shape[1] // 2, shape[2])) < 0.5)  # rows with nothing occupied
        want = (ndimage.distance_transform_edt(1 - occ) * 0.1).astype(np.float32)
        ctx = _lib.default_context()
        try:
            for generic in (0, _lib.NEO_EDT_GENERIC_LINES):
                ctx.check(ctx.lib.neo_esdf_build_config(ctx.h, generic))
                g3 = npa.ESDF3D.from_occupancy(occ, 0.1, (0.0, 0.0, 0.0), layout="linear", want_dist=True)
                assert np.array_equal(g3.dist, want), (shape, generic)
        finally:
            ctx.check(ctx.lib.neo_esdf_build_config(ctx.h, 0))


def test_esdf_build_3d_line_lengths_around_powers_of_two():
    """the y / z passes solve a line by monotone minima over spacings 2^k: line lengths 2 .. 34 on both axes, with and
    without occupied voxels in a line, dense and sparse"""
    from scipy import ndimage
    rng = np.random.default_rng(12)
    for n in (2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 34):
        for shape in ((n, 6, 5), (4, n, 7), (n, n, 3)):
            for dens in (0.02, 0.3):
                occ = (rng.random(shape) < dens).astype(np.uint8)
                occ[0, 0, 0] = 1                                  # at least one occupied voxel
                want = (ndimage.distance_transform_edt(1 - occ) * 0.25).astype(np.float32)
                g3 = npa.ESDF3D.from_occupancy(occ, 0.25, (0.0, 0.0, 0.0), layout="linear", want_dist=True)
                assert np.array_equal(g3.dist, want), (shape, dens)


def test_esdf_build_3d_full_scene_matches_scipy():
    from scipy import ndimage
    occ = synth.occupancy_3d(1, n=160, res=30.0 / 160)
    want = (ndimage.distance_transform_edt(1 - occ) * (30.0 / 160)).astype(np.float32)
    g3 = npa.ESDF3D.from_occupancy(occ, 30.0 / 160, synth.DOMAIN_ORIGIN, want_dist=True)
    assert np.array_equal(g3.dist, want)


def test_trilinear_lookup_matches_oracle_and_ties_back_to_2d():
    d = load(golden("g2_esdf_0.npz")[0])
    res = float(d["res"])
    vol = np.repeat(d["esdf_map"][None].astype(np.float32), 6, axis=0)
    origin = (d["origin"][0], d["origin"][1], 0.0)
    o3 = onp.Grid3DESDF(vol, res, origin)
    rng = np.random.default_rng(1)
    h, w = d["esdf_map"].shape
    pts = rng.uniform([origin[0] - 0.2, origin[1] - 0.2, -0.1], [origin[0] + w * res + 0.2, origin[1] + h * res + 0.2, 0.7], (3000, 3))
    for layout in ("linear", "yz4", "cell8", "brick"):
        for store, tol in (("f32", 1e-12), ("f16", 2e-3)):
            g3 = npa.ESDF3D(vol, res, origin, store=store, layout=layout)
            dis, grd = g3.query(pts)
            od = np.array([o3.lookup(p)[0] for p in pts]); og = np.array([o3.lookup(p)[1] for p in pts])
            assert np.max(np.abs(dis - od)) <= tol
            assert np.max(np.abs(grd - og)) <= tol * 20 + 1e-12
    # at cell centres of a z-constant field, trilinear == nearest-2D
    g3 = npa.ESDF3D(vol, res, origin, store="f32")
    rows = rng.integers(0, h, 500); cols = rng.integers(0, w, 500)
    centres = np.stack([origin[0] + (cols + 0.5) * res, origin[1] + (rows + 0.5) * res, np.full(500, 0.25)], axis=1)
    dis, grd = g3.query(centres)
    assert np.max(np.abs(dis - d["esdf_map"][rows, cols].astype(np.float32).astype(np.float64))) < 1e-12
    assert np.max(np.abs(grd[:, 2])) == 0.0


# ----------------------------------------------------------------------------- one evaluation
@pytest.mark.parametrize("path", golden("g1_eval_s*.npz"))
def test_cost_grad_matches_reference_g1(path):
    d = load(path)
    m = _gpu_map(d)
    # (the all-fp32 mode against the same fixtures: tests/test_gpu_reference_fixtures.py)
    for dtype, tol in (("f64", 1e-10), ("f32", 2e-5)):
        bp = npa.BatchPlanner(sample_dtype=dtype)
        for M in (3, 21, 41):
            t = f"M{M}_"
            out = bp.cost_grad(m, d[t + "x"][None], d[t + "head"][None], d[t + "tail"][None], want_coeffs=True)
            assert rel_err(out["coeffs"][0], d[t + "coeffs"]) < 1e-11
            assert abs(out["cost"][0] - d[t + "cost"]) <= tol * abs(d[t + "cost"])
            assert rel_err(out["costs"][0], d[t + "costs"]) < tol
            assert rel_err(out["grad"][0], d[t + "grad"]) < tol


def _random_requests(rng, B, M, D, map_extent):
    head = np.zeros((B, 3, D)); tail = np.zeros((B, 3, D))
    lo, hi = map_extent
    head[:, 0] = rng.uniform(lo, lo + 0.2 * (hi - lo), (B, D))
    tail[:, 0] = rng.uniform(lo + 0.7 * (hi - lo), hi, (B, D))
    head[:, 1] = rng.normal(0, 0.4, (B, D)); head[:, 2] = rng.normal(0, 0.3, (B, D))
    tail[:, 1] = rng.normal(0, 0.4, (B, D)); tail[:, 2] = rng.normal(0, 0.3, (B, D))
    k = np.arange(1, M)[None, None, :] / M
    wp = head[:, 0, :, None] + (tail[:, 0] - head[:, 0])[:, :, None] * k + rng.normal(0, 0.5, (B, D, M - 1))
    ts = rng.uniform(0.55, 4.8, (B, M))
    return head, tail, wp, ts


@pytest.mark.parametrize("M,D", [(1, 2), (2, 2), (3, 2), (5, 3), (21, 2), (21, 3), (41, 3), (48, 2), (64, 3)])
def test_cost_grad_matches_oracle_on_random_batches(M, D):
    """edge shapes: a single piece (no joint system), two pieces, the maximum 64 pieces"""
    rng = np.random.default_rng(100 * M + D)
    occ = synth.occupancy_2d(M, count=40)
    o2 = onp.GridESDF(occ, 0.1, 300, 300, (0.0, -15.0))
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    B = 6
    lo = np.array([1.0, -6.0, 0.5][:D]); hi = np.array([28.0, 6.0, 3.0][:D])
    head, tail, wp, ts = _random_requests(rng, B, M, D, (lo, hi))
    for stale in (True, False):
        bp = npa.BatchPlanner(sample_dtype="f64", stale_T=stale)
        x = bp.pack_x(wp, ts)
        out = bp.cost_grad(m, x, head, tail, want_coeffs=True)
        for b in range(B):
            pl = onp.OraclePlanner(onp.PlannerParams(), stale_T=stale)
            pl.read_planning_conditions(o2, head[b], tail[b], wp[b], ts[b])
            if M == 1 and stale:
                continue                 # the reference itself fails for M = 1 (unbound T at :529)
            c = pl.get_cost(x[b]); g = pl.get_grad(x[b])
            assert rel_err(out["coeffs"][b], pl.coeffs) < 1e-10
            assert abs(out["cost"][b] - c) <= 1e-10 * abs(c)
            assert rel_err(out["costs"][b], pl.costs) < 1e-10
            assert rel_err(out["grad"][b], g) < 1e-9


def test_cost_grad_trilinear_matches_oracle():
    rng = np.random.default_rng(5)
    n = 48
    occ = np.zeros((n, n, n), np.uint8)
    for _ in range(12):
        a = rng.integers(2, n - 6, 3)
        occ[a[0]:a[0] + rng.integers(2, 6), a[1]:a[1] + rng.integers(2, 6), a[2]:a[2] + rng.integers(2, 6)] = 1
    from scipy import ndimage
    res = 0.25
    dist = (ndimage.distance_transform_edt(1 - occ) * res).astype(np.float32)
    origin = (-1.0, -6.0, 0.0)
    o3 = onp.Grid3DESDF(dist, res, origin)
    for layout in ("linear", "yz4", "cell8", "brick"):
        g3 = npa.ESDF3D(dist, res, origin, store="f32", layout=layout)
        for M, B in ((3, 4), (21, 4), (41, 2)):
            head, tail, wp, ts = _random_requests(rng, B, M, 3, (np.array([0.0, -5.0, 1.0]), np.array([10.5, 5.0, 8.0])))
            ts = rng.uniform(0.7, 3.0, (B, M))
            # "f32x": the all-fp32 mode (solve and adjoint in fp32 too, NEO_FLAG_F32_SOLVE)
            for dtype, tol in (("f64", 1e-10), ("f32", 2e-5), ("f32x", 4e-5)):
                bp = npa.BatchPlanner(sample_dtype=dtype)
                x = bp.pack_x(wp, ts)
                out = bp.cost_grad(g3, x, head, tail)
                for b in range(B):
                    pl = onp.OraclePlanner(onp.PlannerParams())
                    pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
                    c = pl.get_cost(x[b]); g = pl.get_grad(x[b])
                    assert abs(out["cost"][b] - c) <= tol * abs(c)
                    assert rel_err(out["grad"][b], g) < tol * 5


def test_every_layout_holds_the_same_numbers_and_gives_the_same_bits():
    """the field layouts (linear, yz-quads, cell-packed, corner bricks) store the same corner values: every mode's
    evaluation and every whole run is bit-identical across them, for fp32 and fp16 storage, on odd grid sizes (partly
    filled blocks at the upper faces of the brick layout) -- only the addresses differ"""
    rng = np.random.default_rng(77)
    from scipy import ndimage
    for shape in ((31, 45, 38), (8, 9, 7), (2, 3, 5)):
        occ = (rng.random(shape) < 0.004).astype(np.uint8)
        occ[0] = 1
        res = 0.3
        dist = (ndimage.distance_transform_edt(1 - occ) * res).astype(np.float32)
        origin = (0.0, -0.5 * shape[1] * res, 0.0)
        hi = np.array([shape[2] * res, 0.5 * shape[1] * res, shape[0] * res])
        lo = np.array([0.0, -0.5 * shape[1] * res, 0.0])
        for store in ("f32", "f16"):
            maps = {lay: npa.ESDF3D(dist, res, origin, store=store, layout=lay) for lay in ("linear", "yz4", "cell8", "brick")}
            pts = rng.uniform(lo - 0.2, hi + 0.2, (4000, 3))
            ref_d, ref_g = maps["linear"].query(pts)
            for lay in ("yz4", "cell8", "brick"):
                d_, g_ = maps[lay].query(pts)
                assert np.array_equal(d_, ref_d) and np.array_equal(g_, ref_g), (shape, store, lay)
            for M, B in ((3, 16), (21, 8), (41, 4)):
                head, tail, wp, ts = _random_requests(rng, B, M, 3, (lo + 0.3, hi - 0.3))
                ts = rng.uniform(0.6, 2.0, (B, M))
                for dtype in ("f64", "f32", "f32x"):
                    bp = npa.BatchPlanner(sample_dtype=dtype)
                    x = bp.pack_x(wp, ts)
                    ref = bp.cost_grad(maps["linear"], x, head, tail)
                    for lay in ("yz4", "brick"):
                        got = bp.cost_grad(maps[lay], x, head, tail)
                        assert np.array_equal(got["cost"], ref["cost"]) and np.array_equal(got["grad"], ref["grad"]), (shape, store, lay, dtype, M)
                    if M == 21:
                        ro = bp.optimize(maps["yz4"], x, head, tail)
                        rb = bp.optimize(maps["brick"], x, head, tail)
                        assert np.array_equal(ro["x"], rb["x"]) and np.array_equal(ro["nfev"], rb["nfev"]), (shape, store, dtype)


def test_all_fp32_joint_solve_by_cyclic_reduction_for_every_piece_count():
    """The all-fp32 mode solves the joint systems by parallel cyclic reduction (csrc/neo_device.hpp pcr_solve): zero levels
    at M = 2, one at M = 3, ... six at M = 64, lane = (piece, dimension) up to M = 21 and lane = piece beyond.  For every
    piece count the coefficients, the cost and the gradient agree with the fp64 kernels (block Thomas) to fp32 round-off
    of the solve, on durations spread over [T_min, T_max]."""
    rng = np.random.default_rng(23)
    n = 32
    dist = np.full((n, n, n), 4.0, np.float32)
    dist[:, :, :6] = np.linspace(0.0, 1.2, 6)[None, None, :]
    dist[10:14, 12:18, :] = 0.05
    g3 = npa.ESDF3D(dist, 0.4, (0.0, -6.4, 0.0), store="f32", layout="yz4")
    for M in (1, 2, 3, 4, 5, 8, 9, 16, 17, 21, 22, 33, 41, 63, 64):
        B = 3
        head, tail, wp, _ = _random_requests(rng, B, M, 3, (np.array([1.0, -5.0, 1.0]), np.array([11.5, 5.0, 10.0])))
        ts = rng.uniform(0.55, 4.8, (B, M))
        x = npa.BatchPlanner().pack_x(wp, ts)
        ref = npa.BatchPlanner(sample_dtype="f64").cost_grad(g3, x, head, tail, want_coeffs=True)
        got = npa.BatchPlanner(sample_dtype="f32x").cost_grad(g3, x, head, tail, want_coeffs=True)
        assert np.all(got["status"] == 0) and np.all(ref["status"] == 0), M
        for b in range(B):
            assert rel_err(got["coeffs"][b], ref["coeffs"][b]) < (2e-5 if M <= 41 else 1e-4), (M, b, rel_err(got["coeffs"][b], ref["coeffs"][b]))
            assert abs(got["cost"][b] - ref["cost"][b]) <= 1e-4 * abs(ref["cost"][b]), (M, b)
            assert rel_err(got["grad"][b], ref["grad"][b]) < 5e-4, (M, b, rel_err(got["grad"][b], ref["grad"][b]))


def test_sampled_terms_kernel_matches_oracle():
    """the ESDF-lookup kernel alone: add_sampled_cost + add_sampled_grad_CT (:392-466)"""
    d = load(golden("g1_eval_s1.npz")[0])
    m = _gpu_map(d)
    o2 = _oracle_map(d)
    for M in (3, 21, 41):
        t = f"M{M}_"
        pl = onp.OraclePlanner(onp.PlannerParams())
        x = d[t + "x"]
        pl.read_planning_conditions(o2, d[t + "head"], d[t + "tail"], x[:2 * (M - 1)].reshape(2, M - 1), d[t + "ts"])
        pl.get_cost(x)
        pl.reset_cost(); pl.add_sampled_cost()
        pl.reset_grad_CT(); pl.add_sampled_grad_CT()
        for dtype, tol in (("f64", 1e-11), ("f32", 2e-5)):
            out = npa.BatchPlanner(sample_dtype=dtype).sampled_terms(m, pl.coeffs[None], pl.ts[None])
            assert rel_err(out["costs2"][0], pl.costs[2:]) < tol
            assert rel_err(out["grad_C"][0], pl.grad_C) < tol
            assert rel_err(out["grad_T"][0], pl.grad_T) < tol


@pytest.mark.parametrize("M", [1, 2, 3, 7, 21, 41, 64])
def test_sampled_terms_on_ragged_durations(M):
    """the ESDF-lookup kernel on ragged durations: pieces shorter than delta_t (no sample at all -- round 5: their partials were
    left as the caller's buffer had them), pieces with one or two samples, one long piece among short ones, every piece alike,
    a trajectory without any sample -- fp32 sampling against fp64 (pinned to the oracle above) on a continuous 3-D field,
    trajectory by trajectory, and bit-reproducible."""
    import torch
    rng = np.random.default_rng(500 + M)
    dist = synth.esdf_3d(3, n=100, res=0.3, canopy=20)
    g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="brick")
    B, D = 48, 3
    head, tail, wp, ts0 = synth.replan_requests(11, B, max(M - 1, 0), D=3, **synth.VOLUME) if M > 1 else (None,) * 4
    if M == 1:
        head = np.zeros((B, 3, 3)); tail = np.zeros((B, 3, 3))
        head[:, 0] = rng.uniform([1, -10, 1], [5, 10, 20], (B, 3)); tail[:, 0] = head[:, 0] + rng.uniform(5, 12, (B, 3)) * [1, 0.3, 0.2]
        wp = np.zeros((B, 3, 0)); ts0 = np.full((B, 1), 3.0)
    bp64 = npa.BatchPlanner(sample_dtype="f64")
    coeffs = bp64.cost_grad(g3, bp64.pack_x(wp, ts0), head, tail, want_coeffs=True)["coeffs"]
    ts = rng.uniform(0.55, 4.8, (B, M))
    ts[:8] = rng.uniform(0.02, 0.35, (8, M))                       # 0 .. 3 samples a piece
    ts[8:16] = np.where(rng.random((8, M)) < 0.5, 0.05, ts[8:16])      # pieces without samples among ordinary ones
    ts[16:24] = 0.12; ts[16:24, M // 2] = 4.7                          # one long piece among one-sample pieces
    ts[24:28] = 2.5                                                    # every piece alike (a fresh guess)
    ts[28] = 0.03                                                      # a trajectory without any sample
    a = bp64.sampled_terms(g3, coeffs, ts)
    bp32 = npa.BatchPlanner(sample_dtype="f32")
    b = bp32.sampled_terms(g3, coeffs, ts)
    b2 = bp32.sampled_terms(g3, coeffs, ts)
    ns = np.floor(ts / 0.1).astype(int)
    for k in ("costs2", "grad_C", "grad_T"):
        assert np.array_equal(b[k], b2[k]), k
        assert np.isfinite(b[k]).all()
        for t in range(B):          # trajectory by trajectory: a misrouted partial sum does not hide behind another's scale
            # (1e-3: these durations do not belong to the coefficients, so samples sit right at the penalties' thresholds,
            #  where fp32 positions move a cubed difference by 1e-4 relative; a misrouted sum is an error of order one)
            sc = max(np.abs(a[k][t]).max(), 1e-3 * np.abs(a[k]).max())
            assert np.abs(b[k][t] - a[k][t]).max() <= 1e-3 * sc, (k, t, ns[t])
    assert np.all(b["costs2"][28] == 0) and np.all(b["grad_C"][28] == 0) and np.all(b["grad_T"][28] == 0)
    # round 6: fp32 coefficient / partials buffers (neo_sampled_terms_batch_f32): the kernel rounds fp64 coefficients to fp32
    # on entry and its fp32 partials to fp64 on exit, so the fp32 buffers carry the same bits
    c32 = bp32.sampled_terms(g3, coeffs, ts, io32=True)
    assert c32["grad_C"].dtype == np.float32 and c32["grad_T"].dtype == np.float32
    assert np.array_equal(c32["costs2"], b["costs2"])
    assert np.array_equal(c32["grad_C"], b["grad_C"].astype(np.float32)) and np.array_equal(c32["grad_C"].astype(np.float64), b["grad_C"])
    assert np.array_equal(c32["grad_T"].astype(np.float64), b["grad_T"])
    with pytest.raises(Exception):      # fp64 sampling has no fp32-buffer form
        bp64.sampled_terms(g3, coeffs, ts, io32=True)
    # pieces without samples have no partials
    gC = b["grad_C"].reshape(B, M, 6, D)
    assert np.all(gC[ns == 0] == 0) and np.all(b["grad_T"][ns == 0] == 0)
    assert np.all(a["grad_C"].reshape(B, M, 6, D)[ns == 0] == 0) and np.all(a["grad_T"][ns == 0] == 0)


def test_3d_problem_on_z_constant_field_reproduces_2d_costs():
    """D = 3 with v_z = 0 on the 2-D map gives the D = 2 costs"""
    rng = np.random.default_rng(9)
    occ = synth.occupancy_2d(2)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    B, M = 8, 7
    head2, tail2, wp2, ts = _random_requests(rng, B, M, 2, (np.array([1.0, -6.0]), np.array([28.0, 6.0])))
    head3 = np.zeros((B, 3, 3)); tail3 = np.zeros((B, 3, 3)); wp3 = np.full((B, 3, M - 1), 2.0)
    head3[:, :, :2] = head2; tail3[:, :, :2] = tail2; wp3[:, :2] = wp2
    head3[:, 0, 2] = 2.0; tail3[:, 0, 2] = 2.0
    bp = npa.BatchPlanner()
    a = bp.cost_grad(m, bp.pack_x(wp2, ts), head2, tail2)
    b = bp.cost_grad(m, bp.pack_x(wp3, ts), head3, tail3)
    assert rel_err(b["costs"], a["costs"]) < 1e-12


def test_numeric_range_status_like_math_exp():
    d = load(golden("g1_eval_s0.npz")[0])
    m = _gpu_map(d)
    x = d["M3_x"].copy()
    x[-1] = -710.0                                    # math.exp(710) overflows (:481)
    out = npa.BatchPlanner().cost_grad(m, x[None], d["M3_head"][None], d["M3_tail"][None])
    assert out["status"][0] == 4
    # get_grad_T2tau squares a Python float, (1 + math.exp(-tau))**2 (:490): OverflowError already
    # for exp(-tau) > sqrt(DBL_MAX), i.e. tau < -354.89..., although get_cost still succeeds there
    x[-1] = -355.0
    out = npa.BatchPlanner().cost_grad(m, x[None], d["M3_head"][None], d["M3_tail"][None])
    assert out["status"][0] == 4 and np.isfinite(out["cost"][0])
    x[-1] = -354.0
    out = npa.BatchPlanner().cost_grad(m, x[None], d["M3_head"][None], d["M3_tail"][None])
    assert out["status"][0] == 0 and np.isfinite(out["cost"][0]) and np.all(np.isfinite(out["grad"][0]))


def test_all_fp32_mode_keeps_the_fp64_statuses_and_a_finite_gradient_for_runaway_tau():
    """in the all-fp32 mode (Num = float) expf(-tau) overflows from -tau = 88.72 on; T is then T_min exactly
    (as in fp64 to rounding), the gradient must stay finite (no inf / inf), and NUMERIC_RANGE must be raised where the
    fp64 modes raise it (the tests are made on tau: -tau > 354.89 for the gradient, > 709.78 for the cost), not earlier
    and not never.  The optimiser started there must end with a finite result and the same status as the mixed mode."""
    rng = np.random.default_rng(11)
    n = 40
    dist = np.full((n, n, n), 5.0, np.float32)
    dist[:, :, :4] = np.linspace(0.0, 1.0, 4)[None, None, :]
    g3 = npa.ESDF3D(dist, 0.25, (0.0, -5.0, 0.0), store="f32", layout="yz4")
    M, B = 5, 4
    head, tail, wp, ts = _random_requests(rng, B, M, 3, (np.array([1.5, -4.0, 1.0]), np.array([8.5, 4.0, 8.0])))
    for dtype in ("f32x", "f32", "f64"):
        bp = npa.BatchPlanner(sample_dtype=dtype)
        x = bp.pack_x(wp, ts)
        nq = 3 * (M - 1)
        for tau, want in ((-80.0, 0), (-100.0, 0), (-300.0, 0), (-354.0, 0), (-356.0, 4), (-400.0, 4), (-711.0, 4)):
            xx = x.copy()
            xx[:, nq + 1] = tau
            out = bp.cost_grad(g3, xx, head, tail)
            assert np.all(out["status"] == want), (dtype, tau, out["status"])
            if want == 0:
                assert np.all(np.isfinite(out["cost"])) and np.all(np.isfinite(out["grad"])), (dtype, tau)
                # the duration sits on T_min: its tau entry of the gradient is (numerically) zero
                assert np.all(np.abs(out["grad"][:, nq + 1]) <= 1e-20 * np.abs(out["grad"]).max(axis=1)), (dtype, tau)
        xx = x.copy()
        xx[:, nq + 1] = -100.0
        r = bp.optimize(g3, xx, head, tail)
        assert np.all(np.isfinite(r["x"])) and np.all(np.isfinite(r["final_cost"])), dtype
        assert set(np.unique(r["status"])) <= {0, 1, 2}, (dtype, r["status"])


# ----------------------------------------------------------------------------- the optimiser
def _run_entry(pl, d, m):
    entry = str(d["entry"])
    if int(d["np_seed"]) >= 0:
        np.random.seed(int(d["np_seed"]))
    err = ""
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            if entry == "plan":
                pl.plan(m, d["head"], d["tail"])
            elif entry == "batch":
                pl.batch_plan(m, d["head"], d["tail"])
            else:
                pl.read_planning_conditions(m, d["head"], d["tail"], d["init_wpts"], d["init_ts"])
                pl.plan_once()
    except Exception as ex:
        err = f"{type(ex).__name__}:{ex}"
    return err


# Reference runs the device does not follow to the last evaluation (tools/experiments/gpu_g3_status.py on the MI355X; round 5,
# every unit built with -ffp-contract=on).  Both are the SAME replan request recorded twice (seeds 4 and 5 draw the same
# map): same exceptions, same number of L-BFGS-B runs and iterations (10), every evaluation the two runs have in common
# agrees in f to 1e-12 relative, but the LAST line search (iteration 10, f flat to 13 digits between trial steps) takes
# one trial step more or fewer: SciPy 37 / 39 evaluations, the device 36 / 40 -- dcsrch's sufficient-decrease test
# `f <= finit + stp * gtest` decided by the last bits of f -- and ends on the reference's point all the same (finals
# 1e-14 apart).  (Rounds 2 - 4, default contraction: 40 / 35 evaluations, finals 1.3e-4 apart.)
KNOWN_PARTED = {
    "g3_trace_replan_s4.npz": dict(x_rel_max=1e-9, cost_rel_max=1e-9),
    "g3_trace_replan_s5.npz": dict(x_rel_max=1e-9, cost_rel_max=1e-9),
    # a converging M = 21 run of 331 evaluations, far beyond the horizon over which ANY two implementations stay together
    # (the product's own L-BFGS-B on the host, with a bit-identical objective, leaves SciPy's path at evaluation ~134:
    # tests/test_lbfgs_host.py LONG_RUNS).  Asserted: the device follows the reference's recorded evaluations for at least
    # the first 100 (test_parted_runs_first_divergence) and converges to a comparable minimum.  (Round 5: 331 evaluations
    # like the reference by coincidence -- 307 iterations against 308.)
    "g3_trace_once_M21_c0.npz": dict(x_rel_max=1e-1, cost_rel_max=5e-2, long_run=True),
}


def test_planner_reproduces_reference_runs_g3_g5():
    """plan / warm_start_plan retries / batch_plan / plan_once through the reference-shaped class: same exceptions,
    same number of L-BFGS-B runs and iterations, and final control points within north_star's 1e-4 of the reference's
    for EVERY recorded run (in fact 1e-9: they follow SciPy evaluation by evaluation) -- except the runs named in
    KNOWN_PARTED, which are bounded there."""
    import os
    n = n_exact = 0
    for path in golden("g3_trace_*.npz"):
        d = load(path)
        m = _gpu_map(d)
        pl = npa.MinJerkPlanner(npa.PlannerConfig())
        err = _run_entry(pl, d, m)
        assert err.split(":")[0] == str(d["error"]).split(":")[0], path
        known = KNOWN_PARTED.get(os.path.basename(path))
        if not (known or {}).get("long_run"):
            assert pl.iter_num == int(d["iter_num"]), path
        assert pl.opt_running_times == int(d["opt_running_times"]), path
        last = int(d["n_runs"]) - 1
        exact = last < 0 or (pl.last_nfev == int(d[f"r{last}_nfev"]) and pl.iter_num == int(d["iter_num"]))
        n += 1
        n_exact += exact
        if known is None:
            assert exact, (path, pl.last_nfev)
        if exact:
            tol, ctol = 1e-9, 1e-9
        else:       # (a KNOWN_PARTED run; which way its flat tail falls changes with the last bit of any sum)
            tol, ctol = known["x_rel_max"], known["cost_rel_max"]
        assert rel_err(pl.int_wpts, d["final_int_wpts"]) < tol, path
        assert rel_err(pl.ts, d["final_ts"]) < 3 * tol, path
        if "final_cost" in d.files:
            assert abs(pl.final_cost - d["final_cost"]) <= ctol * abs(d["final_cost"]), path
            if "weighted_cost" in d.files and exact and str(d["entry"]) != "batch":
                assert rel_err(pl.weighted_cost, d["weighted_cost"]) < 1e-9
        if "state_cmd_60" in d.files and not (known or {}).get("long_run"):
            hz = int(d["state_cmd_hz"])
            st = pl.get_full_state_cmd(hz)
            assert st.shape == d["state_cmd_60"].shape
            assert rel_err(st, d["state_cmd_60"]) < max(tol, 1e-9) * 10
            assert rel_err(pl.get_pos_array(), d["pos_array"]) < max(tol, 1e-9) * 10
            assert rel_err(pl.get_vel_array(), d["vel_array"]) < max(tol, 1e-9) * 10
    assert n >= 18 and n_exact >= n - len(KNOWN_PARTED), (n_exact, n)


def test_batch_plan_in_one_launch_equals_three_sequential_plan_once():
    """batch_plan optimises its three lateral candidates in ONE launch of three trajectories; the
    result must be what three launches of one give, bit for bit, with the reference's side effects in the same order"""
    for seed in (1, 3, 5):
        occ = synth.occupancy_2d(seed)
        m = npa.ESDF()
        m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
        head = np.array([[0.5, 0.3], [0.4, 0.0]])
        tail = np.array([[5.2, 0.1 * seed], [0.8, 0.0]])
        a = npa.MinJerkPlanner(npa.PlannerConfig())
        with contextlib.redirect_stdout(io.StringIO()) as out_a:
            a.batch_plan(m, head, tail)
        # the same loop with one launch per candidate (expert_planner.py:142-168 as written)
        b = npa.MinJerkPlanner(npa.PlannerConfig())
        cands, ts = b.batch_generate_init_variables(head, tail)
        best_w = np.zeros(cands.shape); best_t = np.zeros((3, len(ts))); cost = np.zeros(3)
        with contextlib.redirect_stdout(io.StringIO()) as out_b:
            for i in range(3):
                try:
                    b.read_planning_conditions(m, head, tail, cands[i], ts)
                    b.plan_once()
                    best_w[i] = b.int_wpts; best_t[i] = b.ts; cost[i] = b.weighted_cost.sum()
                    print(f"batch_cost[{i}] = {cost[i]}")
                except Exception as ex:
                    print(f"The {i}th attempt is deprecated for {ex}")
                    cost[i] = np.inf
                if np.min(cost) < np.inf:
                    k = np.argmin(cost)
                    b.int_wpts = best_w[k]; b.ts = best_t[k]; b.final_cost = cost[k]
        assert np.array_equal(a.int_wpts, b.int_wpts) and np.array_equal(a.ts, b.ts)
        assert a.final_cost == b.final_cost and a.iter_num == b.iter_num and a.opt_running_times == b.opt_running_times
        assert out_a.getvalue() == out_b.getvalue()


def test_lane_groups_on_the_2d_reference_map_match_the_default_kernel():
    """the lane-group kernel (eight replans per wavefront) on the reference's own shape -- D = 2,
    M = 3 (n = 7), nearest-cell 2-D map -- in both arithmetics.  A group strides a piece's samples over fewer lanes than
    the default kernel, so sums associate differently: in fp64 nearly every run is the default kernel's run to the last
    digits, and the batch statistics are the same; bit-reproducible; independent of the batch neighbours."""
    occ = synth.occupancy_2d(2)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    B, M = 1024, 3
    head, tail, wp, ts = synth.replan_requests(21, B, M - 1, D=2, length_range=(4.0, 6.0), jitter=0.3)
    for dtype, frac_same, xtol in (("f64", 0.9, 1e-7), ("f32", 0.4, 1e-3)):
        bd = npa.BatchPlanner(sample_dtype=dtype)
        bg = npa.BatchPlanner(sample_dtype=dtype, lane_groups=True)
        x0 = bd.pack_x(wp, ts)
        rd = bd.optimize(m, x0, head, tail)
        rg = bg.optimize(m, x0, head, tail)
        rg2 = bg.optimize(m, x0, head, tail)
        assert np.array_equal(rg["x"], rg2["x"]) and np.array_equal(rg["nfev"], rg2["nfev"])
        assert set(np.unique(rg["status"])) <= {0, 1, 2, 3, 4, 5}
        same = (rd["nfev"] == rg["nfev"]) & (rd["status"] == rg["status"])
        assert same.mean() >= frac_same, (dtype, same.mean())
        dx = np.abs(rd["x"][same] - rg["x"][same]).max(axis=1) / np.abs(rd["x"][same]).max(axis=1)
        assert np.quantile(dx, 0.9) <= xtol, (dtype, np.quantile(dx, 0.9))
        ok = (rd["status"] <= 2) & (rg["status"] <= 2)
        assert abs(np.median(rd["final_cost"][ok]) - np.median(rg["final_cost"][ok])) <= 1e-3 * np.median(rd["final_cost"][ok])
        assert abs(rd["nfev"].mean() - rg["nfev"].mean()) <= 0.03 * rd["nfev"].mean()
        pick = np.array([0, 5, 77, 600, 1023])
        r3 = bg.optimize(m, x0[pick], head[pick], tail[pick])
        assert np.array_equal(r3["x"], rg["x"][pick])
    # and against SciPy on the oracle objective, as test_optimize_batch_matches_cpu_optimizer does for the default kernel
    o2 = onp.GridESDF(occ, 0.1, 300, 300, (0.0, -15.0))
    rg = npa.BatchPlanner(lane_groups=True).optimize(m, npa.BatchPlanner().pack_x(wp, ts), head, tail)
    wq, tq = npa.BatchPlanner().unpack_x(rg["x"], M, 2)
    n_exact = 0
    for b in range(24):
        pl, err = _oracle_plan_once(o2, head[b], tail[b], wp[b], ts[b])
        if err == "overflow":
            continue
        n_exact += rg["nfev"][b] == pl.last_result.nfev and rel_err(wq[b], pl.int_wpts) < 1e-7
    assert n_exact >= 18, n_exact


def test_parted_runs_first_divergence():
    """for every recorded reference run the device does not follow to its last evaluation
    (KNOWN_PARTED), the LAST L-BFGS-B run is traced on the device (neo_optimize_trace_xg) and laid beside SciPy's
    recorded evaluations: printed per run -- the first evaluation at which the evaluated points differ by more than 1e-7
    -- and asserted: every evaluation before it agrees in f to 1e-9, and the first divergence is late (the parted flat
    tails of the replan scenarios; evaluation >= 100 of 331 for the long converging run)."""
    import ctypes
    import os
    from neo_planner_amd import _lib
    for name, known in KNOWN_PARTED.items():
        d = load(golden(name)[0])
        m = _gpu_map(d)
        last = int(d["n_runs"]) - 1
        x0 = d[f"r{last}_x0"]
        ref_x, ref_f = d[f"r{last}_eval_x"], d[f"r{last}_eval_f"]
        M = (len(x0) + 2) // 3
        hs = np.zeros((1, 3, 2)); tl = np.zeros((1, 3, 2))
        hs[0, :d["head"].shape[0]] = d["head"]; tl[0, :d["tail"].shape[0]] = d["tail"]
        import torch
        dev = torch.device("cuda", 0)
        ctx = m.ctx
        cap = 512
        tr = torch.zeros(1, cap, 4, dtype=torch.float64, device=dev)
        xg = torch.zeros(1, cap, 2, len(x0), dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), cap))
        ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, ctypes.c_void_p(xg.data_ptr()), cap))
        try:
            r = npa.BatchPlanner(ctx=ctx).optimize(m, x0[None], hs, tl)
        finally:
            ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
            ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, None, 0))
        E = int(r["nfev"][0])
        gx, gf = xg[0, :E, 0].cpu().numpy(), tr[0, :E, 0].cpu().numpy()
        # SciPy records fun(x) calls; the point evaluated again from its cache does not appear twice there either
        k = min(E, len(ref_x))
        scale = np.maximum(np.abs(ref_x[:k]).max(axis=1), 1.0)
        err = np.abs(gx[:k] - ref_x[:k]).max(axis=1) / scale
        bad = np.flatnonzero(err > 1e-7)
        first = int(bad[0]) if len(bad) else k
        upto = min(first, 100)       # (a long run: the points themselves drift apart by then, and f with them)
        ferr = np.abs(gf[:upto] - ref_f[:upto]) / np.abs(ref_f[:upto])
        print(f"{name}: device {E} evaluations, SciPy {len(ref_x)}; evaluated points agree to 1e-7 up to evaluation {first} "
              f"(max f deviation before it {ferr.max():.1e}); point deviation at 25/50/75/100 % of the common run: "
              f"{[float(err[int(q * (k - 1))]) for q in (0.25, 0.5, 0.75, 1.0)]}")
        assert first >= (100 if known.get("long_run") else 0.8 * len(ref_x)), (name, first)
        assert ferr.max() <= (1e-6 if known.get("long_run") else 1e-9), (name, ferr.max())


def _oracle_plan_once(o_map, head, tail, wp, ts):
    pl = onp.OraclePlanner(onp.PlannerParams())
    pl.read_planning_conditions(o_map, head, tail, wp, ts)
    err = ""
    try:
        pl.plan_once()
    except ValueError:
        err = "collision"
    except OverflowError:
        err = "overflow"
    return pl, err


@pytest.mark.parametrize("M,B", [(3, 48), (21, 12)])
def test_optimize_batch_matches_cpu_optimizer(M, B):
    """the batched device optimiser against SciPy L-BFGS-B on the oracle objective, same inputs"""
    seed = 11
    occ = synth.occupancy_2d(seed)
    o2 = onp.GridESDF(occ, 0.1, 300, 300, (0.0, -15.0))
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    lr = (4.0, 6.0) if M == 3 else (10.0, 28.0)
    head, tail, wp, ts = synth.replan_requests(seed, B, M - 1, D=2, length_range=lr, jitter=0.3)
    bp = npa.BatchPlanner()
    res = bp.optimize(m, bp.pack_x(wp, ts), head, tail)
    wq, tq = bp.unpack_x(res["x"], M, 2)
    n_exact = 0
    gpu_costs, cpu_costs = [], []
    for b in range(B):
        pl, err = _oracle_plan_once(o2, head[b], tail[b], wp[b], ts[b])
        if err == "overflow":
            assert res["status"][b] in (4, 5)
            continue
        r = pl.last_result
        exact = int(res["nit"][b]) == r.nit and int(res["nfev"][b]) == r.nfev
        n_exact += exact
        gpu_cost = (res["costs_last"][b] * pl.weights).sum()
        gpu_costs.append(gpu_cost)
        cpu_costs.append(pl.final_cost)
        if exact:
            # the run followed SciPy step for step: the BASELINE.json bar (1e-4) holds with margin
            assert rel_err(wq[b], pl.int_wpts) < 1e-8, b
            assert rel_err(tq[b], pl.ts) < 1e-8
            assert bool(res["collision"][b]) == (err == "collision")
            assert abs(gpu_cost - pl.final_cost) <= 1e-9 * abs(pl.final_cost)
        else:
            # a line-search decision fell the other way on a jump of the objective:
            # both are valid L-BFGS-B runs that may settle in different local minima; the device's must not
            # be noticeably worse, and must be of the same size (seen: 13 % better on one M = 21 case)
            assert gpu_cost <= 1.1 * pl.final_cost, (b, gpu_cost, pl.final_cost)
            assert gpu_cost >= 0.5 * pl.final_cost, (b, gpu_cost, pl.final_cost)
    # short runs (M = 3, ~20 evaluations) almost always stay on SciPy's path; M = 21 runs take ~100
    # evaluations with several failed searches each, and about half of them meet a flipped decision
    assert n_exact >= (0.75 if M == 3 else 0.4) * B, (n_exact, B)
    assert abs(np.mean(gpu_costs) - np.mean(cpu_costs)) <= 0.02 * abs(np.mean(cpu_costs))


# ----------------------------------------------------------------------------- full-size properties
@pytest.fixture(scope="module")
def cfg2():
    """BASELINE.json configs[1]: 4096 trajectories, 20 waypoints, one 300^3 fp32 field"""
    g3 = npa.ESDF3D.from_occupancy(synth.occupancy_3d(0), synth.RES, synth.DOMAIN_ORIGIN, store="f32")
    head, tail, wp, ts = synth.replan_requests(0, 4096, 20, D=3)
    return g3, head, tail, wp, ts


@pytest.mark.parametrize("dtype", ["f64", "f32", "f32x"])
def test_full_size_batch_properties(cfg2, dtype):
    g3, head, tail, wp, ts = cfg2
    bp = npa.BatchPlanner(sample_dtype=dtype)
    x0 = bp.pack_x(wp, ts)
    e0 = bp.cost_grad(g3, x0, head, tail)
    r1 = bp.optimize(g3, x0, head, tail)
    r2 = bp.optimize(g3, x0, head, tail)
    # bit-reproducible (no atomics anywhere on the path)
    assert np.array_equal(r1["x"], r2["x"]) and np.array_equal(r1["nfev"], r2["nfev"])
    assert set(np.unique(r1["status"])) <= {0, 1, 2, 3, 4, 5}
    ok = r1["status"] <= 2
    assert ok.mean() > 0.95
    # L-BFGS-B never accepts an increase: final objective <= initial objective
    assert np.all(r1["final_cost"][ok] <= e0["cost"][ok] * (1 + 1e-12))
    # costs reported for the final x are what one more evaluation at that x gives
    e1 = bp.cost_grad(g3, r1["x"], head, tail)
    assert rel_err(e1["costs"][ok], r1["costs"][ok]) < (1e-12 if dtype == "f64" else 1e-5)
    # a trajectory's result does not depend on its batch neighbours, nor on the dispatch order
    pick = np.array([0, 1, 777, 2048, 4095])
    r3 = bp.optimize(g3, x0[pick], head[pick], tail[pick])
    assert np.array_equal(r3["x"], r1["x"][pick])
    r4 = bp.optimize(g3, x0, head, tail, order=False)
    assert np.array_equal(r4["x"], r1["x"]) and np.array_equal(r4["nfev"], r1["nfev"])
    perm = bp.expected_effort_order(head, tail, ts)
    assert sorted(perm.tolist()) == list(range(4096))
    # every iteration count is sane and evaluations >= iterations
    assert np.all(r1["nfev"] >= r1["nit"]) and r1["nfev"].max() < 15000


def test_multi_scene_batch_uses_the_right_map():
    occ_a, occ_b = synth.occupancy_2d(1), synth.occupancy_2d(2, count=40)
    ma, mb = npa.ESDF(), npa.ESDF()
    ma.occupancy_map_cb(synth.OccupancyGridMsg(occ_a))
    mb.occupancy_map_cb(synth.OccupancyGridMsg(occ_b))
    head, tail, wp, ts = synth.replan_requests(5, 16, 2, D=2, length_range=(4.0, 6.0), jitter=0.3)
    bp = npa.BatchPlanner()
    x0 = bp.pack_x(wp, ts)
    ra = bp.optimize(ma, x0, head, tail)
    rb = bp.optimize(mb, x0, head, tail)
    ids = np.where(np.arange(16) % 2 == 0, ma.scene_id, mb.scene_id).astype(np.int32)
    rm = bp.optimize(ma, x0, head, tail, scene_ids=ids)
    want = np.where((np.arange(16) % 2 == 0)[:, None], ra["x"], rb["x"])
    assert np.array_equal(rm["x"], want)


def _cpu_run_3d(args):
    """worker (spawned, never touches the GPU): one trajectory through the CPU oracle on the 3-D field"""
    scene, n, b = args
    import numpy as _np
    from neo_planner_amd import synth as _synth
    from oracle import minco_np as _onp
    res = 30.0 / n
    o3 = _onp.Grid3DESDF(_synth.esdf_3d(scene, n=n, res=res), res, _synth.DOMAIN_ORIGIN)
    head, tail, wp, ts = _synth.replan_requests(scene, 64, 20, D=3)
    pl = _onp.OraclePlanner(_onp.PlannerParams())
    pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
    try:
        pl.plan_once()
    except Exception:
        pass
    r = pl.last_result
    return b, (r.nit, r.nfev) if r is not None else (-1, -1), float(_np.dot(pl.costs, pl.weights)), pl.int_wpts.copy()


def test_3d_runs_follow_the_cpu_optimizer_in_fp64_mode():
    """north-star shape (M = 21, D = 3, trilinear field) in parity mode: most runs take exactly the CPU's
    evaluations and end at the CPU's control points; all of them end at comparable cost"""
    import multiprocessing as mp
    scene, n, B = 3, 100, 24
    with mp.get_context("spawn").Pool(min(B, 24)) as pool:
        cpu = sorted(pool.map(_cpu_run_3d, [(scene, n, b) for b in range(B)]))
    res = 30.0 / n
    g3 = npa.ESDF3D.from_occupancy(synth.occupancy_3d(scene, n=n, res=res), res, synth.DOMAIN_ORIGIN)
    head, tail, wp, ts = synth.replan_requests(scene, 64, 20, D=3)
    bp = npa.BatchPlanner(sample_dtype="f64")
    out = bp.optimize(g3, bp.pack_x(wp[:B], ts[:B]), head[:B], tail[:B])
    wq, _ = bp.unpack_x(out["x"], 21, 3)
    w = np.array(bp.cfg.weights)
    same = 0
    rels = []
    for b, (nit, nfev), cost, wpts in cpu:
        if nit < 0:
            continue
        gpu_cost = float((out["costs_last"][b] * w).sum())
        rels.append(abs(gpu_cost - cost) / abs(cost))
        if int(out["nit"][b]) == nit and int(out["nfev"][b]) == nfev and rel_err(wq[b], wpts) < 1e-6:
            same += 1                      # the run followed the CPU step for step
            assert rels[-1] < 1e-6
    assert same >= 0.4 * len(rels), (same, len(rels))
    # runs that left the CPU's path (a decision on a jump of the objective fell the other way) settle in
    # other local minima: comparable cost, no tighter statement holds for them
    assert np.median(rels) < 1e-4 and np.percentile(rels, 80) < 0.02 and max(rels) < 0.5, \
        (np.median(rels), np.percentile(rels, 80), max(rels))


def _cpu_run_3d_small(args):
    """worker: one M = 3 trajectory through the CPU oracle on the 3-D field"""
    scene, n, b, B = args
    import numpy as _np
    from neo_planner_amd import synth as _synth
    from oracle import minco_np as _onp
    res = 30.0 / n
    o3 = _onp.Grid3DESDF(_synth.esdf_3d(scene, n=n, res=res), res, _synth.DOMAIN_ORIGIN)
    head, tail, wp, ts = _synth.replan_requests(scene, B, 2, D=3, length_range=(4.0, 6.0))
    pl = _onp.OraclePlanner(_onp.PlannerParams())
    pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
    try:
        pl.plan_once()
    except Exception:
        pass
    r = pl.last_result
    return b, (r.nit, r.nfev) if r is not None else (-1, -1), float(_np.dot(pl.costs, pl.weights))


def test_lane_group_kernel_against_the_cpu_optimizer():
    """the eight-trajectories-per-wavefront kernel (M = 3, fp32 sampling) against SciPy L-BFGS-B on the oracle
    objective, same inputs: the bar of the fp32 mode (final cost, §3 of DESIGN.md)"""
    import multiprocessing as mp
    scene, n, B = 5, 100, 32
    with mp.get_context("spawn").Pool(8) as pool:
        cpu = sorted(pool.map(_cpu_run_3d_small, [(scene, n, b, B) for b in range(B)]))
    res = 30.0 / n
    g3 = npa.ESDF3D(synth.esdf_3d(scene, n=n, res=res), res, synth.DOMAIN_ORIGIN, store="f32")
    head, tail, wp, ts = synth.replan_requests(scene, B, 2, D=3, length_range=(4.0, 6.0))
    bp = npa.BatchPlanner(sample_dtype="f32", lane_groups=True)
    out = bp.optimize(g3, bp.pack_x(wp, ts), head, tail)
    w = np.array(bp.cfg.weights)
    rels, same = [], 0
    for b, (nit, nfev), cost in cpu:
        if nit < 0 or out["status"][b] > 2:
            continue
        gpu_cost = float((out["costs_last"][b] * w).sum())
        rels.append(abs(gpu_cost - cost) / abs(cost))
        same += int(out["nit"][b]) == nit and int(out["nfev"][b]) == nfev
    assert len(rels) >= 0.8 * B
    assert np.median(rels) < 1e-4 and np.percentile(rels, 80) < 2e-2, (np.median(rels), np.percentile(rels, 80))
    assert same >= 0.2 * len(rels), (same, len(rels))


def test_every_boundary_row_reaches_every_optimiser_kernel():
    """Head velocity / acceleration and tail velocity / acceleration all non-zero (the synthetic replan requests leave
    three of those rows at zero).  The optimiser kernels copy the boundary states once per trajectory (LDS copy in the
    lane = (piece, dimension) layout and in the lane-group kernel, scalar loads where the lane holds all dimensions):
    the first evaluation of a run must be the evaluation kernel's value at the start point, and the lane-group kernel
    must end where the one-trajectory-per-wavefront kernel ends."""
    import ctypes
    import torch
    from neo_planner_amd import _lib
    rng = np.random.default_rng(77)
    n = 32
    dist = np.full((n, n, n), 4.0, np.float32)
    dist[:, :, :6] = np.linspace(0.0, 1.2, 6)[None, None, :]
    dist[10:14, 12:18, :] = 0.05
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    g3 = npa.ESDF3D(dist, 0.4, (0.0, -6.4, 0.0), store="f32", layout="yz4", ctx=ctx)
    cap = 4
    # (piece counts on both sides of every kernel switch: lane = (piece, dimension) up to 21, FLAT slots 1 / 2 / 4 at
    #  n = 64 / 128, the five-level cap of the fp32 reduction from 34)
    for M in (1, 2, 3, 16, 17, 21, 22, 25, 32, 33, 34, 41, 64):
        B = 8
        head, tail, wp, _ = _random_requests(rng, B, M, 3, (np.array([1.0, -5.0, 1.0]), np.array([11.5, 5.0, 10.0])))
        ts = rng.uniform(0.8, 2.5, (B, M))
        nv = 3 * (M - 1) + M
        for dtype, tol in (("f64", 1e-12), ("f32", 2e-5), ("f32x", 1e-4)):
            for waves in ((None, 2) if dtype != "f64" else (None,)):
                bp = npa.BatchPlanner(ctx=ctx, sample_dtype=dtype, waves_per_simd=waves)
                bp._sync()
                x0 = bp.pack_x(wp, ts)
                want = bp.cost_grad(g3, x0, head, tail)["cost"]
                t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                d_x0, d_x = t(x0), torch.empty(B, nv, dtype=torch.float64, device=dev)
                costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
                nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
                tr = torch.zeros(B, cap, 4, dtype=torch.float64, device=dev)
                ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), cap))
                try:
                    bp.optimize_dev(g3, d_x, t(head), t(tail), costs, last, nit, nfev, st, x0=d_x0)
                    ctx.synchronize()
                finally:
                    ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
                f0 = tr[:, 0, 0].cpu().numpy()
                assert np.all(np.abs(f0 - want) <= tol * np.abs(want)), (M, dtype, waves, np.abs(f0 - want) / np.abs(want))
    # lane groups (M = 3): fp64 against the default fp64 kernel, the all-fp32 groups against the all-fp32 default
    M, B = 3, 64
    head, tail, wp, _ = _random_requests(rng, B, M, 3, (np.array([1.0, -5.0, 1.0]), np.array([11.5, 5.0, 10.0])))
    ts = rng.uniform(0.8, 2.5, (B, M))
    # (the fp64 pair pins the indexing -- the kernels are one template; the all-fp32 pair runs hard problems in fp32 and
    #  is held to the spread two fp32 runs of them show, which a wrong row would still exceed everywhere)
    # (f32x: the two kernels sum a piece's samples in different orders -- lanes per piece differ -- and a tenth of these
    #  problems is chaotic enough to end somewhere else entirely; the median is the discriminating figure)
    for dtype, tol_med, tol_90 in (("f64", 1e-9, 1e-3), ("f32x", 2e-2, 2.0)):
        a = npa.BatchPlanner(ctx=ctx, sample_dtype=dtype)
        g = npa.BatchPlanner(ctx=ctx, sample_dtype=dtype, lane_groups=True)
        x0 = a.pack_x(wp, ts)
        ra, rg = a.optimize(g3, x0, head, tail), g.optimize(g3, x0, head, tail)
        # (random boundary states of this size make hard problems: a fifth of the runs leave the range of exp(-tau) --
        #  status 4 -- in either kernel; the comparison is over the runs both finish)
        ok = (ra["status"] <= 2) & (rg["status"] <= 2)
        assert ok.sum() >= 0.5 * B and np.mean(ra["status"] == rg["status"]) >= (0.9 if dtype == "f64" else 0.7), \
            (dtype, ok.sum(), np.mean(ra["status"] == rg["status"]))
        rel = np.abs(rg["final_cost"][ok] - ra["final_cost"][ok]) / np.abs(ra["final_cost"][ok])
        assert np.median(rel) < tol_med and np.percentile(rel, 90) < tol_90, (dtype, np.median(rel), np.percentile(rel, 90))
