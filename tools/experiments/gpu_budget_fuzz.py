#!/usr/bin/env python3
"""Budgeted launches against unbudgeted ones on whole cfg2 batches (4096 requests, 300^3 brick field), every arithmetic mode,
random budgets: every trajectory's x, cost terms, counts and status must be equal bit for bit.  Prints one line per case."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
grid, B, M = 300, 4096, 21
res = 30.0 / grid
occ = synth.occupancy_3d(1, n=grid, res=res, canopy=80)
g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
rng = np.random.default_rng(5)
bad = 0
for seed in (1, 1001):
    head, tail, wp, ts = synth.replan_requests(seed, B, M - 1, D=3, **synth.VOLUME)
    for mode in ("f32x", "f32", "f64"):
        bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode)
        x0 = bp.pack_x(wp, ts)
        ref = bp.optimize(g3, x0, head, tail)
        for budget in sorted(int(v) for v in rng.integers(5, 400, 3)):
            got = bp.optimize_budgeted(g3, x0, head, tail, budget)
            same = all(np.array_equal(got[k], ref[k]) for k in ("x", "costs", "costs_last", "nit", "nfev", "status", "collision"))
            bad += not same
            print(f"requests {seed} mode {mode} budget {budget}: launches {got['launch_sizes'][:6]}{'...' if len(got['launch_sizes']) > 6 else ''} "
                  f"({len(got['launch_sizes'])}), max nfev {int(ref['nfev'].max())}, equal: {same}", flush=True)
print("mismatching cases:", bad)
