#!/usr/bin/env python3
"""per-case deviations of the all-fp32 mode from the G1 reference evaluations on the 2-D nearest-cell map, next to the
reference objective's own jump under a perturbation of x of fp32 size (is the point at a cell face?)"""
import glob, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import neo_planner_amd as npa
from neo_planner_amd import synth
from oracle import minco_np as onp

for path in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "g1_eval_s*.npz"))):
    d = np.load(path)
    m = npa.ESDF(); m.occupancy_map_cb(synth.OccupancyGridMsg(d["occ"], float(d["res"]), d["origin"]))
    o2 = onp.GridESDF(d["occ"], float(d["res"]), d["occ"].shape[1], d["occ"].shape[0], d["origin"])
    for M in (3, 21, 41):
        t = f"M{M}_"
        x = d[t + "x"]
        row = []
        for mode in ("f32", "f32x"):
            r = npa.BatchPlanner(sample_dtype=mode).cost_grad(m, x[None], d[t + "head"][None], d[t + "tail"][None], want_coeffs=True)
            row.append((abs(r["cost"][0] - float(d[t + "cost"])) / abs(float(d[t + "cost"])),
                        np.abs(r["costs"][0] - d[t + "costs"]) / np.maximum(np.abs(d[t + "costs"]), 1e-300),
                        np.abs(r["coeffs"][0] - d[t + "coeffs"]).max()))
        pl = onp.OraclePlanner(onp.PlannerParams())
        pl.read_planning_conditions(o2, d[t + "head"], d[t + "tail"], x[:2 * (M - 1)].reshape(2, M - 1), np.ones(M))
        c0 = pl.get_cost(x)
        rng = np.random.default_rng(M)
        jump = 0.0
        for k in range(12):
            xp = x * (1.0 + 4e-6 * rng.standard_normal(x.shape))
            jump = max(jump, abs(pl.get_cost(xp) - c0) / abs(c0))
        print(f"{os.path.basename(path)} M={M:2d} f32 cost {row[0][0]:.1e} | f32x cost {row[1][0]:.1e} terms {np.array2string(row[1][1], precision=1)} "
              f"coeff abs err {row[1][2]:.1e} | reference cost jump under 4e-6 relative noise on x: {jump:.1e}", flush=True)
