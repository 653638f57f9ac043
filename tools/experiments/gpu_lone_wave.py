#!/usr/bin/env python3
"""Latency of the all-fp32 optimiser kernel when wavefronts have SIMDs to themselves: cfg2 batch 0 in launches of 4096 and of
256 trajectories, default allocation (three wavefronts per SIMD, one sample a lane in flight) against NEO_FLAG_ONE_WAVE_PER_SIMD
(experiment library built with -DNEO_X_ONE_WAVE: 512 registers, four samples in flight).  Prints ms per launch, the longest
run's evaluations and microseconds per evaluation of that run; checks that both give the same bits."""
import ctypes, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
grid, M, D = 300, 21, 3
res = 30.0 / grid
occ = synth.occupancy_3d(0, n=grid, res=res, canopy=80)
g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
head, tail, wp, ts = synth.replan_requests(0, 4096, M - 1, D=D, **synth.VOLUME)
ref = {}
for B in (4096, 256):
    for waves in (2, 1):
        bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x", waves_per_simd=waves)
        bp._sync()        # (optimize_dev does not push the planner's parameters itself)
        x0 = torch.from_numpy(bp.pack_x(wp[:B], ts[:B])).to(dev)
        h = torch.from_numpy(head[:B]).to(dev); tl = torch.from_numpy(tail[:B]).to(dev)
        x = torch.empty_like(x0); c = torch.zeros(B, 4, dtype=torch.float64, device=dev); l = torch.zeros_like(c)
        nit = torch.zeros(B, dtype=torch.int32, device=dev); nf = torch.zeros_like(nit); s = torch.zeros_like(nit)
        order = torch.from_numpy(bp.expected_effort_order(head[:B], tail[:B], ts[:B])).to(dev)
        ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(order.data_ptr()), B))
        run = lambda: bp.optimize_dev(g3, x, h, tl, c, l, nit, nf, s, x0=x0)
        for _ in range(3):
            run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize(); ms = 1e2 * (time.perf_counter() - t0)
        mx = int(nf.max().item())
        key = (B,)
        same = None
        if key in ref:
            same = bool(torch.equal(ref[key][0], x) and torch.equal(ref[key][1], nf))
        else:
            ref[key] = (x.clone(), nf.clone())
        print(f"B {B} waves_per_simd {waves}: {ms:.3f} ms a launch, {B / ms:.0f} k traj/s, longest run {mx} evaluations -> "
              f"{1e3 * ms / mx:.2f} us per evaluation of the longest run, same bits as the default: {same}", flush=True)
