#!/usr/bin/env python3
"""cfg5-shaped optimiser runs (M = 41, 600^3 fp16 field) in every arithmetic mode and kernel variant: median final cost,
status histogram, mean evaluations -- a quick health check of the n = 161 kernels (NEO_PLANNER_LIB picks the library)."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib
n, M = 600, 41
res = 30.0 / n
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
occ = synth.occupancy_3d(3, n=n, res=res, canopy=80)
head, tail, wp, ts = synth.replan_requests(3, 4096, M - 1, D=3, **synth.VOLUME)
for layout in (sys.argv[1:] or ["linear", "brick"]):
    g16 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), res, synth.DOMAIN_ORIGIN, store="f16", layout=layout, ctx=ctx)
    for mode, B in ((("f64", 24), ("f64", 2048)) if os.environ.get("NEO_REPRO_F64") else (("f64", 24), ("f64", 2048), ("f32", 24), ("f32", 4096), ("f32x", 24), ("f32x", 4096))):
        bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode)
        x0 = bp.pack_x(wp, ts)
        r = bp.optimize(g16, x0[:B], head[:B], tail[:B])
        e0 = bp.cost_grad(g16, x0[:B], head[:B], tail[:B])
        print(layout, mode, B, "median final", float(np.median(r["final_cost"])), "median initial", float(np.median(e0["cost"])),
              "status", np.bincount(r["status"], minlength=7).tolist(), "mean nfev", float(r["nfev"].mean()), flush=True)
    ctx.check(ctx.lib.neo_esdf_drop(ctx.h, g16.scene_id))
