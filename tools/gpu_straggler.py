#!/usr/bin/env python3
"""Where do the longest runs of optimize_kernel spend their time?  Needs a library built with
NEO_BUILD_DEFS=-DNEO_STAMPS (the sample-counter buffer then carries 8 values per trajectory)."""
import ctypes, os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth

grid = 300; res = 30.0 / grid
# NEO_PLANAR=1: the round-1 workload (z = 2 m, no canopy); NEO_LAYOUT, NEO_WAVES (1 | 2): layout / register allocation
planar = bool(os.environ.get("NEO_PLANAR"))
dist = synth.esdf_3d(0, n=grid, res=res, canopy=0 if planar else 80)
B, M, D = 4096, int(os.environ.get("NEO_M", "21")), 3     # NEO_M: pieces (41 = cfg5's shape)
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=D, **({} if planar else synth.VOLUME))
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx,
                layout=os.environ.get("NEO_LAYOUT", "yz4"))
bp = npa.BatchPlanner(ctx=ctx, sample_dtype=os.environ.get("NEO_DTYPE", "f32x"), waves_per_simd=int(os.environ.get("NEO_WAVES", "2")))
bp.flags |= int(os.environ.get("NEO_FLAGS_OR", "0"))      # e.g. 1024: gathers dropped by the range check (timing only)
bp._sync()
x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev); x = x0.clone()
h = torch.from_numpy(head).to(dev); tl = torch.from_numpy(tail).to(dev)
costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); status = torch.zeros_like(nit)
cnt = torch.zeros(B, 8, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.neo_optimize_sample_counter(ctx.h, ctypes.c_void_p(cnt.data_ptr())))
order = torch.from_numpy(bp.expected_effort_order(head, tail, ts).astype(np.int32)).to(dev)
ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(order.data_ptr()), B))
for _ in range(2):
    x.copy_(x0)
    bp.optimize_dev(g3, x, h, tl, costs, last, nit, nfev, status)
    torch.cuda.synchronize()
c = cnt.cpu().numpy().astype(np.float64)
ev = c[:, 1]; tick = 0.01  # us per tick (100 MHz)
tot = c[:, 5] * tick
start = (c[:, 6] - c[:, 6].min()) * tick
print(f"launch: last finish {np.max(start + tot):.0f} us; mean run {tot.mean():.0f} us; evaluations mean {ev.mean():.1f} max {ev.max():.0f}")
def row(sel, name):
    e = ev[sel].sum()
    f, s_, b_ = c[sel, 2].sum() * tick / e, c[sel, 3].sum() * tick / e, c[sel, 4].sum() * tick / e
    t = tot[sel].sum() / e
    tl_ = c[sel, 7].sum() * tick / e
    print(f"{name:<28} n={sel.sum():5d}  per evaluation: total {t:6.2f} us = forward {f:5.2f} + sample {s_:5.2f} + backward {b_:5.2f} + optimiser {t - f - s_ - b_:5.2f} (of which two-loop {tl_:5.2f});"
          f"  samples/eval {c[sel, 0].sum() / e:6.1f}")
row(np.ones(B, bool), "all runs")
idx = np.argsort(-tot)
for k in (1, 8, 64):
    sel = np.zeros(B, bool); sel[idx[:k]] = True
    row(sel, f"longest {k}")
    if k == 8:
        for i in idx[:8]:
            print(f"    traj {i}: start {start[i]:7.0f} us, duration {tot[i]:7.0f} us, {int(ev[i])} evaluations, status {int(status[i]) & 0xff}")
late = start + tot
sel = np.zeros(B, bool); sel[np.argsort(-late)[:8]] = True
row(sel, "last 8 to finish")
