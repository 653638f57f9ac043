#!/usr/bin/env python3
"""Where do the evaluation kernel and the optimiser kernel part, bit for bit, in the all-fp32 mode?  (cost terms: forward +
sampling; gradient by index: waypoints / durations)"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
dist = synth.esdf_3d(5, n=120, res=0.25, canopy=30)
g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), 0.25, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
B, M, D, CAP = 16, 21, 3, 400
n = D * (M - 1) + M
head, tail, wp, ts = synth.replan_requests(11, B, M - 1, D=3, **synth.VOLUME)
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
bp._sync()
x0 = bp.pack_x(wp, ts)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_x0, d_x, d_h, d_t = t(x0), torch.empty(B, n, dtype=torch.float64, device=dev), t(head), t(tail)
costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
xg = torch.zeros(B, CAP, 2, n, dtype=torch.float64, device=dev)
tr = torch.zeros(B, CAP, 4, dtype=torch.float64, device=dev)
ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, ctypes.c_void_p(xg.data_ptr()), CAP))
ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), CAP))
bp.optimize_dev(g3, d_x, d_h, d_t, costs, last, nit, nfev, st, x0=d_x0)
ctx.synchronize()
ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, None, 0)); ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
nf = nfev.cpu().numpy(); xgh = xg.cpu().numpy(); trh = tr.cpu().numpy()
for b in range(0, B, 4):
    E = min(int(nf[b]), CAP)
    pts = xgh[b, :E, 0]
    e = bp.cost_grad(g3, pts, np.repeat(head[b:b + 1], E, axis=0), np.repeat(tail[b:b + 1], E, axis=0))
    f_eq = np.mean(e["cost"] == trh[b, :E, 0]) if "cost" in e else None
    gd = e["grad"] != xgh[b, :E, 1]
    print(f"b={b} E={E} keys={list(e.keys())} f equal share {f_eq}; grad entries differing: waypoint part {gd[:, :n - M].mean():.3f}, duration part {gd[:, n - M:].mean():.3f}; "
          f"max rel diff {np.max(np.abs(e['grad'] - xgh[b, :E, 1]) / (np.abs(e['grad']).max(axis=1, keepdims=True))):.2e}")
