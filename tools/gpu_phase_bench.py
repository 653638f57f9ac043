#!/usr/bin/env python3
"""timing experiments on a GPU box: per-evaluation cost of the phases of eval_kernel at cfg2
(the phase switches exist only in -DNEO_EXPERIMENTS builds: NEO_BUILD_DEFS=-DNEO_EXPERIMENTS NEO_BUILD_OUT=<lib> python -m
neo_planner_amd.build, then NEO_PLANNER_LIB=<lib>; with the product library every row times the full kernel)"""
import ctypes, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib

grid = 300
res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
B, M, D = 4096, 21, 3
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=D)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
n = D * (M - 1) + M
for dtype in ("f32", "f64"):
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype=dtype)
    x = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    h = torch.from_numpy(head).to(dev); tl = torch.from_numpy(tail).to(dev)
    cost = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
    g = torch.zeros(B, n, dtype=torch.float64, device=dev); stt = torch.zeros(B, dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for dbg, name in ((0, "full"), (1, "no sample loop"), (2, "no joint sweeps"), (3, "neither")):
        bp._sync(); ctx.set_params(flags=dbg)
        def run():
            ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, p(x), p(h), p(tl), p(cost), p(c4), p(g), None, p(stt)))
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 50
        for _ in range(R): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
        print(f"{dtype} eval_kernel {name:18s}: {dt*1e6:8.1f} us per launch of {B} evaluations  ({dt/B*1e9:.1f} ns/eval)")
    ctx.set_params(flags=0)

print("--- latency / scaling of eval_kernel and optimize_kernel with batch size (f32)")
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32"); bp._sync(); ctx.set_params(flags=0)
for Bs in (1, 64, 256, 1024, 2048, 4096):
    def run():
        ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, Bs, M, D, p(x), p(h), p(tl), p(cost), p(c4), p(g), None, p(stt)))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 50
    for _ in range(R): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    print(f"eval_kernel B={Bs:5d}: {dt*1e6:8.1f} us per launch")
x0 = x.clone()
costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); status = torch.zeros_like(nit)
for Bs in (1, 64, 1024, 4096):
    def run():
        x.copy_(x0)
        ctx.check(ctx.lib.neo_optimize_batch_dev(ctx.h, g3.scene_id, None, Bs, M, D, p(x), p(h), p(tl), p(costs), p(last), p(nit), p(nfev), p(status)))
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 5
    for _ in range(R): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    nf = nfev[:Bs].cpu().numpy()
    print(f"optimize_kernel B={Bs:5d}: {dt*1e3:8.3f} ms; nfev sum {nf.sum()} max {nf.max()} -> {dt/nf.max()*1e6:.1f} us per eval of the longest run; "
          f"{dt/nf.sum()*1e9:.1f} ns per eval amortised")

print("--- single-wave latency split (B=1, f32)")
for dbg, name in ((0, "full"), (1, "no sample loop"), (2, "no joint sweeps"), (8, "no factor sweep"), (16, "no scans"), (3, "neither")):
    bp._sync(); ctx.set_params(flags=dbg)
    def run():
        ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, 1, M, D, p(x0), p(h), p(tl), p(cost), p(c4), p(g), None, p(stt)))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 200
    for _ in range(R): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    print(f"B=1 eval_kernel {name:18s}: {dt*1e6:8.1f} us")
ctx.set_params(flags=0)
