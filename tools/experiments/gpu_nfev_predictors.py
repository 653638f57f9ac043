#!/usr/bin/env python3
"""What predicts the length of a run?  cfg2 request batches: evaluations per run against features known before the launch
(path length, the four cost terms at the start point).  Spearman rank correlations + how much of the top-1 % longest runs a
predictor's top decile catches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
from scipy.stats import spearmanr
import neo_planner_amd as npa
from neo_planner_amd import synth
grid, B, M = 300, 4096, 21
res = 30.0 / grid
occ = synth.occupancy_3d(0, n=grid, res=res, canopy=80)
g3 = npa.ESDF3D.from_occupancy(occ, res, synth.DOMAIN_ORIGIN, store="f32", layout="brick")
bp = npa.BatchPlanner(sample_dtype="f32x")
for seed in (0, 1, 2):
    h, t, w, ts = synth.replan_requests(seed, B, M - 1, D=3, **synth.VOLUME)
    x0 = bp.pack_x(w, ts)
    e = bp.cost_grad(g3, x0, h, t)
    o = bp.optimize(g3, x0, h, t)
    nfev = o["nfev"].astype(float)
    L = np.linalg.norm(t[:, 0, :] - h[:, 0, :], axis=1)
    gn = np.linalg.norm(e["grad"], axis=1)
    feats = {"length": L, "cost0": e["cost"], "energy0": e["costs"][:, 0], "feas0": e["costs"][:, 2], "coll0": e["costs"][:, 3], "|grad0|": gn}
    top = nfev >= np.quantile(nfev, 0.99)
    print(f"seed {seed}: mean nfev {nfev.mean():.1f} max {nfev.max():.0f}")
    for k, v in feats.items():
        rho = spearmanr(v, nfev).correlation
        dec = v >= np.quantile(v, 0.9)
        print(f"   {k:8s} spearman {rho:+.3f}   top-1% longest runs inside the predictor's top decile: {float((top & dec).sum()) / top.sum():.2f}")
