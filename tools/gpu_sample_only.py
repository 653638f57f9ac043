#!/usr/bin/env python3
"""The ESDF-lookup kernel alone on bench.py's cfg2 batch 0 at the initial guess (brick layout, spatial dispatch order):
mean launch duration from HIP events, for the in-tree library or the one NEO_PLANNER_LIB names.  Small enough to sit
under `rocprofv3 --pmc ... -- python3 tools/gpu_sample_only.py` (counters of sample_kernel@4096 / @163840).

    python3 tools/gpu_sample_only.py [--whole] [--reps N] [--f64io]
Since round 6 the coefficient / partials buffers are fp32 (neo_sampled_terms_batch_f32_dev; --f64io: the fp64 buffers of
neo_sampled_terms_batch_dev) and two durations are printed: one launch per HIP-event pair (the context's profile scope: what
bench.py reported up to round 5, ~2 us of event overhead at this size) and 20 launches back to back between one pair.
"""
import ctypes
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32x")
if "--flags" in sys.argv:     # comparison switches of include/neo_planner.h
    bp.flags |= int(sys.argv[sys.argv.index("--flags") + 1])
bp._sync()
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
io_dt = torch.float64 if "--f64io" in sys.argv else torch.float32
coeffs = coeffs.to(io_dt)
c2 = torch.zeros(Ba, 2, dtype=torch.float64, device=dev); gC = torch.zeros(Ba, 6 * M, D, dtype=io_dt, device=dev)
gT = torch.zeros(Ba, M, dtype=io_dt, device=dev)
fn = ctx.lib.neo_sampled_terms_batch_dev if "--f64io" in sys.argv else ctx.lib.neo_sampled_terms_batch_f32_dev
run = lambda: ctx.check(fn(ctx.h, g3.scene_id, Ba, M, D, pp(coeffs), pp(d_ts), pp(c2), pp(gC), pp(gT)))
# warm-up long enough for the clocks to settle (a cold chip runs the first launches a third slower), then the best of
# five blocks of `reps` launches
t_end = time.time() + (0.0 if "--no-warmup" in sys.argv else 1.5)
while time.time() < t_end:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
for _ in range(3):
    run()
torch.cuda.synchronize()
best = 1e30
best_k = 1e30
K = 20
nl, ms = ctypes.c_int64(), ctypes.c_double()
for _ in range(1 if "--no-warmup" in sys.argv else 5):
    ctx.check(ctx.lib.neo_profile_reset(ctx.h)); ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
    ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(nl), ctypes.byref(ms)))
    best = min(best, 1e3 * ms.value / max(nl.value, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(K):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    best_k = min(best_k, 1e3 * e0.elapsed_time(e1) / K)
us = best
ns = int(np.floor(ts_a / bp.cfg.delta_t).astype(np.int64).sum())
by = ns * 32.0 + Ba * (2 * n * 4 + 20)
print(f"lib {os.environ.get('NEO_PLANNER_LIB', 'in-tree')} ({'fp64' if '--f64io' in sys.argv else 'fp32'} buffers): sample_kernel@{Ba} one launch per event pair "
      f"{us:.2f} us, frac_8d2 {by / (us * 1e-6) / 8e12:.4f}; {K} back to back {best_k:.2f} us, frac_8d2 {by / (best_k * 1e-6) / 8e12:.4f}; "
      f"checksum {float(gC.sum()):.9e} {float(c2.sum()):.9e}")
