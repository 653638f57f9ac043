#!/usr/bin/env python3
"""experiment: the ESDF sample kernel on whole trajectories (M=21, L=3 lanes per piece) versus the same
samples presented as half/third trajectories (more lanes per piece, more waves)"""
import ctypes, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth
grid = 300; res = 30.0 / grid
B, M, D = 4096, 21, 3
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=D)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
g3 = npa.ESDF3D.from_occupancy(synth.occupancy_3d(0), res, synth.DOMAIN_ORIGIN, ctx=ctx)
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32"); bp._sync()
e0 = bp.cost_grad(g3, bp.pack_x(wp, ts), head, tail, want_coeffs=True)
co = e0["coeffs"].reshape(B, M, 6, D)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for parts in (1, 2, 3, 7):
    Mp = M // parts
    c2 = np.ascontiguousarray(co[:, :Mp * parts].reshape(B * parts, Mp * 6, D)); t2 = np.ascontiguousarray(ts[:, :Mp * parts].reshape(B * parts, Mp))
    dc = torch.from_numpy(c2).to(dev); dt = torch.from_numpy(t2).to(dev)
    c2o = torch.zeros(B * parts, 2, dtype=torch.float64, device=dev); gC = torch.zeros_like(dc); gT = torch.zeros_like(dt)
    run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, B * parts, Mp, D, p(dc), p(dt), p(c2o), p(gC), p(gT)))
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): run()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 100 * 1e6
    print(f"parts {parts}: M={Mp} L={64 // Mp} waves={B * parts}: {us:.1f} us per launch; collision cost sum {c2o[:,1].sum().item():.6f}")
