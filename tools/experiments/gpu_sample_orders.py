#!/usr/bin/env python3
"""The ESDF-lookup kernel alone on bench.py's cfg2 batch 0 at the initial guess (brick layout) under different dispatch
orders (sort keys of BatchPlanner.spatial_order): launch duration from HIP events.  Small enough to sit
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

c2 = torch.zeros(Ba, 2, dtype=torch.float64, device=dev); gC = torch.zeros(Ba, 6 * M, D, dtype=torch.float64, device=dev)
gT = torch.zeros(Ba, M, dtype=torch.float64, device=dev)
run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, Ba, M, D, pp(coeffs), pp(d_ts), pp(c2), pp(gC), pp(gT)))
H, T = np.concatenate(hl), np.concatenate(tl)


def morton(q):
    q = q - q.min(axis=0)
    bits = max(1, int(np.ceil(np.log2(max(int(q.max()) + 1, 2)))))
    key = np.zeros(len(q), dtype=np.int64)
    for b in range(bits):
        for d in range(q.shape[1]):
            key |= ((q[:, d] >> b) & 1) << (q.shape[1] * b + d)
    return key


def keys():
    h, t = H[:, 0], T[:, 0]
    mid = 0.5 * (h + t)
    dirv = (t - h) / np.linalg.norm(t - h, axis=1, keepdims=True)
    fl = lambda a, c: np.floor(a / c).astype(np.int64)
    yield "default (3-D midpoint, 1 m)", None, None
    for c in (0.5, 2.0, 4.0):
        yield f"3-D midpoint, {c} m", morton(fl(mid, c)), None
    yield "midpoint (y, z) only, 1 m", morton(fl(mid[:, 1:], 1.0)), None
    yield "start (y, z) + goal (y, z), 2 m", morton(np.concatenate([fl(h[:, 1:], 2.0), fl(t[:, 1:], 2.0)], axis=1)), None
    yield "goal (x, y, z), 2 m", morton(fl(t, 2.0)), None
    yield "heading (8 x 8 bins) then midpoint (y, z) 2 m", (fl(dirv[:, 1] + 1, 0.25) * 8 + fl(dirv[:, 2] + 1, 0.25)) * (1 << 20) + morton(fl(mid[:, 1:], 2.0)), None
    yield "default key, runs of 128", None, 128
    yield "default key, runs of 64", None, 64
    yield "index order", "index", None


# warm-up
order0 = torch.from_numpy(npa.BatchPlanner.spatial_order(H, T)).to(dev)
ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(order0), 1, Ba))
t_end = time.time() + 1.5
while time.time() < t_end:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
nl, ms = ctypes.c_int64(), ctypes.c_double()
ns = int(np.floor(ts_a / bp.cfg.delta_t).astype(np.int64).sum())
by = ns * 32.0 + Ba * (2 * n * 4 + 20)
for name, key, chunk in keys():
    if isinstance(key, str):
        ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, None, 1, 0))
    else:
        od = torch.from_numpy(npa.BatchPlanner.spatial_order(H, T, key=key, chunk=chunk)).to(dev)
        ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(od), 1, Ba))
    best = 1e30
    for _ in range(5):
        ctx.check(ctx.lib.neo_profile_reset(ctx.h)); ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
        ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(nl), ctypes.byref(ms)))
        best = min(best, 1e3 * ms.value / max(nl.value, 1))
    print(f"{name:48s} sample_kernel@{Ba} {best:8.2f} us  frac_8d2 {by / (best * 1e-6) / 8e12:.4f}", flush=True)
