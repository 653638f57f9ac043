#!/usr/bin/env python3
"""a few launches of the stand-alone ESDF sample kernel on the cfg2 batch (for rocprofv3 counter passes)"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
grid = int(os.environ.get("NEO_GRID", "300")); res = 30.0 / grid
layout = os.environ.get("NEO_LAYOUT", "linear")
dist = synth.esdf_3d(0, n=grid, res=res)
B, M = 4096, 21
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=3)
g3 = npa.ESDF3D(dist, res, synth.DOMAIN_ORIGIN, store="f32", layout=layout)
bp = npa.BatchPlanner(sample_dtype="f32")
x0 = bp.pack_x(wp, ts)
e0 = bp.cost_grad(g3, x0, head, tail, want_coeffs=True)
for _ in range(4):
    out = bp.sampled_terms(g3, e0["coeffs"], ts)
print("ok", out["costs2"].sum())
