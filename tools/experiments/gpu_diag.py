#!/usr/bin/env python3
"""ad-hoc diagnostics on a GPU box (not part of the test suite)"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
from oracle import minco_np as onp

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 300
res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
B, M = 4096, 21
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=3)
g3 = npa.ESDF3D(dist, res, synth.DOMAIN_ORIGIN, store="f32")
for dtype in ("f64", "f32"):
    bp = npa.BatchPlanner(sample_dtype=dtype)
    x0 = bp.pack_x(wp, ts)
    e0 = bp.cost_grad(g3, x0, head, tail)
    t0 = time.time(); r = bp.optimize(g3, x0, head, tail); t1 = time.time()
    print(f"{dtype}: grid {grid} optimize {1e3*(t1-t0):.1f} ms nfev mean {r['nfev'].mean():.1f} nit mean {r['nit'].mean():.2f} "
          f"status {np.bincount(r['status'], minlength=6)} cost0 mean {e0['cost'].mean():.4g} median {np.median(e0['cost']):.4g} "
          f"final median {np.median(r['final_cost']):.4g} costs0 mean {e0['costs'].mean(axis=0)}")
    print("   nit hist", np.bincount(np.minimum(r['nit'], 20)))
    print("   first 6: nit", r['nit'][:6], "nfev", r['nfev'][:6], "status", r['status'][:6], "cost0", e0['cost'][:6], "final", r['final_cost'][:6])
o3 = onp.Grid3DESDF(dist, res, synth.DOMAIN_ORIGIN)
for b in range(3):
    pl = onp.OraclePlanner(onp.PlannerParams())
    pl.read_planning_conditions(o3, head[b], tail[b], wp[b], ts[b])
    try:
        pl.plan_once(); e = ""
    except Exception as ex:
        e = str(ex)
    rr = pl.last_result
    print(f"oracle b={b}: nit {rr.nit} nfev {rr.nfev} msg {rr.message} cost {np.dot(pl.costs, pl.weights):.6g} {e}")
