#!/usr/bin/env python3
"""Quick diagnostic run on a GPU box: prints error figures of the HIP path against the oracle and
the golden fixtures (no assertions; the pytest suite in tests/ is the gate).  Usage:
    python tools/gpu_check.py [eval] [esdf] [opt] [tri] [speed]"""
import glob
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))

import numpy as np

import neo_planner_amd as npa
from neo_planner_amd import synth
from oracle import minco_np as onp

G = os.path.join(REPO, "tests", "golden")


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def check_eval():
    print("== per-evaluation parity vs golden G1 (fp64 sampling, then fp32)")
    for dtype in ("f64", "f32"):
        worst = dict(cost=0, costs=0, grad=0, coeffs=0)
        for path in sorted(glob.glob(os.path.join(G, "g1_eval_s*.npz"))):
            d = np.load(path)
            occ = d["occ"]
            m = npa.ESDF()
            m.occupancy_map_cb(synth.OccupancyGridMsg(occ, float(d["res"]), d["origin"]))
            bp = npa.BatchPlanner(sample_dtype=dtype)
            for M in (3, 21, 41):
                t = f"M{M}_"
                out = bp.cost_grad(m, d[t + "x"][None], d[t + "head"][None], d[t + "tail"][None], want_coeffs=True)
                e = dict(cost=abs(out["cost"][0] - d[t + "cost"]) / abs(d[t + "cost"]),
                         costs=rel(out["costs"][0], d[t + "costs"]), grad=rel(out["grad"][0], d[t + "grad"]),
                         coeffs=rel(out["coeffs"][0], d[t + "coeffs"]))
                for k in worst:
                    worst[k] = max(worst[k], e[k])
                if max(e.values()) > (1e-8 if dtype == "f64" else 1e-2):
                    print("   ", os.path.basename(path), "M", M, {k: f"{v:.2e}" for k, v in e.items()})
        print(f"  sample_dtype {dtype}: worst rel err", {k: f"{v:.2e}" for k, v in worst.items()})


def check_esdf():
    print("== ESDF build vs golden G2")
    for path in sorted(glob.glob(os.path.join(G, "g2_esdf_*.npz"))):
        d = np.load(path)
        m = npa.ESDF()
        m.occupancy_map_cb(synth.OccupancyGridMsg(d["occ"], float(d["res"]), d["origin"]))
        dis, grd = m.query(d["probe_pts"])
        print("  ", os.path.basename(path), "esdf equal", np.array_equal(m.esdf_map, d["esdf_map"]),
              "gx", np.array_equal(m.esdf_grad_x, d["esdf_grad_x"]), "gy", np.array_equal(m.esdf_grad_y, d["esdf_grad_y"]),
              "probe dis", np.array_equal(dis, d["probe_dis"]), "probe grad", np.array_equal(grd, d["probe_grad"]),
              "max|d esdf|", float(np.abs(m.esdf_map - d["esdf_map"]).max()))


def check_opt():
    print("== optimiser vs golden G3 traces (through MinJerkPlanner)")
    for path in sorted(glob.glob(os.path.join(G, "g3_trace_*.npz"))):
        d = np.load(path)
        m = npa.ESDF()
        m.occupancy_map_cb(synth.OccupancyGridMsg(d["occ"], float(d["res"]), d["origin"]))
        pl = npa.MinJerkPlanner(npa.PlannerConfig())
        entry = str(d["entry"])
        if int(d["np_seed"]) >= 0:
            np.random.seed(int(d["np_seed"]))
        err = ""
        import io, contextlib
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                if entry == "plan":
                    pl.plan(m, d["head"], d["tail"])
                elif entry == "batch":
                    pl.batch_plan(m, d["head"], d["tail"])
                else:
                    pl.read_planning_conditions(m, d["head"], d["tail"], d["init_wpts"], d["init_ts"])
                    pl.plan_once()
        except Exception as ex:
            err = f"{type(ex).__name__}:{ex}"
        fc = getattr(pl, "final_cost", float("nan"))
        rfc = float(d["final_cost"]) if "final_cost" in d.files else float("nan")
        print(f"  {os.path.basename(path):32s} err '{err}' (ref '{str(d['error'])}') iter_num {pl.iter_num} "
              f"(ref {int(d['iter_num'])}) runs {pl.opt_running_times} (ref {int(d['opt_running_times'])}) "
              f"x {rel(pl.int_wpts, d['final_int_wpts']):.1e} ts {rel(pl.ts, d['final_ts']):.1e} "
              f"cost {fc:.8g} (ref {rfc:.8g})")


def check_tri():
    print("== trilinear 3-D mode vs oracle Grid3DESDF (n = 48^3 synthetic field)")
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
    for layout in ("linear", "brick4"):
        g3 = npa.ESDF3D(dist, res, origin, store="f32", layout=layout)
        pts = rng.uniform([-1.5, -6.5, -0.5], [11.5, 6.5, 12.5], (2000, 3))
        dd, gg = g3.query(pts)
        od = np.array([o3.lookup(p)[0] for p in pts]); og = np.array([o3.lookup(p)[1] for p in pts])
        print(f"  layout {layout}: query dist err {np.abs(dd - od).max():.2e} grad err {np.abs(gg - og).max():.2e}")
        for M, B in ((3, 8), (21, 8), (41, 4)):
            head = np.zeros((B, 3, 3)); tail = np.zeros((B, 3, 3))
            head[:, 0] = rng.uniform([0, -5, 1], [1, 5, 8], (B, 3)); tail[:, 0] = rng.uniform([9, -5, 1], [10.5, 5, 8], (B, 3))
            head[:, 1] = rng.normal(0, 0.3, (B, 3))
            k = np.arange(1, M)[None, None, :] / M
            wp = head[:, 0, :, None] + (tail[:, 0] - head[:, 0])[:, :, None] * k + rng.normal(0, 0.4, (B, 3, M - 1))
            ts = rng.uniform(0.7, 3.0, (B, M))
            for dtype in ("f64", "f32"):
                bp = npa.BatchPlanner(sample_dtype=dtype)
                x = bp.pack_x(wp, ts)
                out = bp.cost_grad(g3, x, head, tail, want_coeffs=True)
                worst = dict(cost=0, grad=0, coeffs=0)
                for b in range(B):
                    pl = onp.OraclePlanner(onp.PlannerParams())
                    pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
                    c = pl.get_cost(x[b]); g = pl.get_grad(x[b])
                    worst["cost"] = max(worst["cost"], abs(out["cost"][b] - c) / abs(c))
                    worst["grad"] = max(worst["grad"], rel(out["grad"][b], g))
                    worst["coeffs"] = max(worst["coeffs"], rel(out["coeffs"][b], pl.coeffs))
                print(f"    M {M} {dtype}: ", {k2: f"{v:.2e}" for k2, v in worst.items()})


def check_speed():
    import torch
    print("== speed: cfg2-like (B=4096, M=21, D=3, 128^3 field for a quick look)")
    n = 128
    dist = synth.esdf_3d(0, n=n, res=30.0 / n)
    g3 = npa.ESDF3D(dist, 30.0 / n, synth.DOMAIN_ORIGIN, store="f32")
    B, M = 4096, 21
    head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=3)
    for dtype in ("f64", "f32"):
        bp = npa.BatchPlanner(sample_dtype=dtype)
        x0 = bp.pack_x(wp, ts)
        t0 = time.time(); out = bp.cost_grad(g3, x0, head, tail); t1 = time.time()
        out = bp.cost_grad(g3, x0, head, tail); t2 = time.time()
        print(f"  {dtype}: cost_grad host-to-host {1e3 * (t2 - t1):.2f} ms (first {1e3 * (t1 - t0):.1f})")
        t0 = time.time(); res = bp.optimize(g3, x0, head, tail); t1 = time.time()
        res = bp.optimize(g3, x0, head, tail); t2 = time.time()
        print(f"  {dtype}: optimize {1e3 * (t2 - t1):.1f} ms -> {B / (t2 - t1):.0f} traj/s; nfev mean {res['nfev'].mean():.1f} "
              f"max {res['nfev'].max()} nit mean {res['nit'].mean():.1f}; status hist {np.bincount(res['status'], minlength=6)}; "
              f"collision {res['collision'].mean():.2f}; cost0 {out['cost'].mean():.4g} -> {res['final_cost'].mean():.4g}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["esdf", "eval", "tri", "opt", "speed"]
    for w in which:
        t0 = time.time()
        try:
            globals()["check_" + w]()
        except Exception as ex:
            import traceback
            traceback.print_exc()
        print(f"   [{w}: {time.time() - t0:.1f} s]", flush=True)
