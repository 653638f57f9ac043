#!/usr/bin/env python3
"""The ESDF-lookup kernel alone on bench.py's cfg2 batch 0 at the initial guess (brick layout, spatial dispatch order):
mean launch duration from HIP events, for the in-tree library or the one NEO_PLANNER_LIB names.  Small enough to sit
under `rocprofv3 --pmc ... -- python3 tools/gpu_sample_only.py` (counters of sample_kernel@4096 / @163840).

    python3 tools/gpu_sample_only.py [--whole] [--reps N]
"""
import ctypes
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib

reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
nb = 40 if "--whole" in sys.argv else 1
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
grid, B, M, D = 300, 4096, 21, 3
res = 30.0 / grid
occ = synth.occupancy_3d(0, n=grid, res=res, canopy=80)
g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", layout="brick", ctx=ctx)
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x"); bp._sync()
pp = lambda t: ctypes.c_void_p(t.data_ptr())
n = D * (M - 1) + M
co, tsl, hl, tl = [], [], [], []
for r in range(nb):
    head, tail, wp, ts = synth.replan_requests(1000 * r, B, M - 1, D=D, **synth.VOLUME)
    x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
    c = torch.zeros(B, 6 * M, D, dtype=torch.float64, device=dev)
    cost = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
    g = torch.zeros(B, n, dtype=torch.float64, device=dev); s1 = torch.zeros(B, dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(x0), pp(torch.from_numpy(head).to(dev)),
                                              pp(torch.from_numpy(tail).to(dev)), pp(cost), pp(c4), pp(g), pp(c), pp(s1)))
    co.append(c); tsl.append(ts); hl.append(head); tl.append(tail)
torch.cuda.synchronize()
coeffs = torch.cat(co); ts_a = np.concatenate(tsl); d_ts = torch.from_numpy(np.ascontiguousarray(ts_a)).to(dev)
Ba = B * nb
order = torch.from_numpy(npa.BatchPlanner.spatial_order(np.concatenate(hl), np.concatenate(tl))).to(dev)
ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(order), 1, Ba))
c2 = torch.zeros(Ba, 2, dtype=torch.float64, device=dev); gC = torch.zeros(Ba, 6 * M, D, dtype=torch.float64, device=dev)
gT = torch.zeros(Ba, M, dtype=torch.float64, device=dev)
run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, Ba, M, D, pp(coeffs), pp(d_ts), pp(c2), pp(gC), pp(gT)))

for _ in range(200):
    run()
torch.cuda.synchronize()
import collections
for trial in range(3):
    run(); torch.cuda.synchronize()
    g = gT.cpu().numpy()
    k0, k1, xcc = g[:, M - 2], g[:, M - 1], g[:, M - 3].astype(int)
    t0 = k0.min()
    s_us, e_us = (k0 - t0) / 100.0, (k1 - t0) / 100.0
    life = e_us - s_us
    q = lambda a: " ".join(f"{v:6.2f}" for v in np.quantile(a, [0, 0.1, 0.5, 0.9, 0.99, 1.0]))
    print(f"trial {trial}: kernel span {e_us.max():.2f} us | start (min p10 p50 p90 p99 max) {q(s_us)} | end {q(e_us)} | life {q(life)}")
    od = order.cpu().numpy()
    slot_of = np.empty(Ba, dtype=np.int64); slot_of[od] = np.arange(Ba)
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"    xcc {x}: {int(m.sum()):5d} waves, slots mod 8 = {sorted(collections.Counter((slot_of[m] % 8).tolist()).items())[:3]}, start p50 {np.median(s_us[m]):5.2f} end p50 {np.median(e_us[m]):5.2f} max {e_us[m].max():5.2f} life p50 {np.median(life[m]):5.2f}")
    late = np.argsort(-e_us)[:8]
    print("    last to end: slots", slot_of[late].tolist(), "start", np.round(s_us[late], 2).tolist(), "life", np.round(life[late], 2).tolist())
