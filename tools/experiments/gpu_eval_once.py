#!/usr/bin/env python3
"""a few launches of eval_kernel (cfg2 batch) and one optimize launch, for rocprofv3 counter runs"""
import ctypes, os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
grid = int(os.environ.get("NEO_GRID", "300")); res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
B, M = 4096, 21
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=3)
g3 = npa.ESDF3D(dist, res, synth.DOMAIN_ORIGIN, store="f32")
bp = npa.BatchPlanner(sample_dtype="f32")
x0 = bp.pack_x(wp, ts)
for _ in range(3):
    e0 = bp.cost_grad(g3, x0, head, tail)
r = bp.optimize(g3, x0, head, tail)
print("evals", int(r["nfev"].sum()))
