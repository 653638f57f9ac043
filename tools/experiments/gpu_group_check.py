#!/usr/bin/env python3
"""lane-group kernel (four trajectories per wavefront) against the default kernel on an M = 3 batch"""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
grid = 300; res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
g3 = npa.ESDF3D(dist, res, synth.DOMAIN_ORIGIN, store="f32")
for B in (7, 256, 4096):
    M = 3
    head, tail, wp, ts = synth.replan_requests(3, B, M - 1, D=3, length_range=(4.0, 6.0))
    outs = []
    for lg in (False, True):
        bp = npa.BatchPlanner(sample_dtype="f32", lane_groups=lg)
        t0 = time.perf_counter()
        outs.append(bp.optimize(g3, bp.pack_x(wp, ts), head, tail, order=False))
        print(f"B={B} lane_groups={lg}: {time.perf_counter() - t0:.3f} s, nfev mean {outs[-1]['nfev'].mean():.2f}, status hist {np.bincount(outs[-1]['status'], minlength=6)}")
    a, b = outs
    same = (a["nfev"] == b["nfev"]) & (a["nit"] == b["nit"]) & (a["status"] == b["status"])
    rel = np.abs(a["final_cost"] - b["final_cost"]) / np.maximum(np.abs(a["final_cost"]), 1e-12)
    dx = np.abs(a["x"] - b["x"]).max(axis=1)
    print(f"   same (nit, nfev, status): {same.mean():.3f}; final cost rel diff median {np.median(rel):.2e} max {rel.max():.2e}; "
          f"max |dx| over runs on the same path: {dx[same].max() if same.any() else float('nan'):.2e}")
