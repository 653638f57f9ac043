#!/usr/bin/env python3
"""Do consecutive cfg2 batches overlap when they are issued on two HIP streams (two contexts)?  The tail of a
launch leaves most SIMDs idle (a few long runs); the next batch can fill them."""
import ctypes, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth

grid = 300; res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
B, M, D = 4096, 21, 3
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=D)
dev = torch.device("cuda", 0)
dist_d = torch.from_numpy(dist).to(dev)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lanes = []
for s in range(NS):
    st = torch.cuda.Stream()
    ctx = npa.Context(0, stream=st.cuda_stream)
    g3 = npa.ESDF3D(dist_d, res, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32"); bp._sync()
    with torch.cuda.stream(st):
        x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev); x = x0.clone()
        h = torch.from_numpy(head).to(dev); tl = torch.from_numpy(tail).to(dev)
        costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
        nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); status = torch.zeros_like(nit)
        order = torch.from_numpy(bp.expected_effort_order(head, tail, ts).astype(np.int32)).to(dev)
    ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(order.data_ptr()), B))
    lanes.append(dict(st=st, ctx=ctx, g3=g3, bp=bp, x0=x0, x=x, h=h, tl=tl, costs=costs, last=last, nit=nit, nfev=nfev, status=status, order=order))
torch.cuda.synchronize()

def step(l):
    with torch.cuda.stream(l["st"]):
        l["x"].copy_(l["x0"])
        l["bp"].optimize_dev(l["g3"], l["x"], l["h"], l["tl"], l["costs"], l["last"], l["nit"], l["nfev"], l["status"])

for k in range(2 * NS):
    step(lanes[k % NS])
torch.cuda.synchronize()
for K in (10, 20):
    t0 = time.perf_counter()
    for k in range(K):
        step(lanes[k % NS])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{NS} stream(s), {K} batches: {dt / K * 1e3:.2f} ms per batch, {B * K / dt:.0f} traj/s; nfev mean {float(lanes[0]['nfev'].float().mean()):.1f}")
