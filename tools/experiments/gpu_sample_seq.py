#!/usr/bin/env python3
"""(Needs the experiment library of tools/probe/seq/README.md: the product has no sequence body and no flags bit 8192.)
Round 6: the ESDF-lookup kernel's sequence body (neo_sample_seq.hpp) against the round-4 body (flags bit 8192) and the fp64
kernel on bench.py's cfg2 batch 0 at the initial guess (brick layout, spatial dispatch order): agreement trajectory by
trajectory, bit reproducibility, independence of the dispatch order, and launch durations -- one launch per HIP-event pair
(what bench.py's `esdf_kernel` reported up to round 5) and K launches back to back between ONE pair.

    python3 tools/experiments/gpu_sample_seq.py [--whole] [--reps N] [--optimised]
--optimised: coefficients / durations of the OPTIMISED batch (converged trajectories: many samples at the velocity bound)
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
    x0 = bp.pack_x(wp, ts)
    if "--optimised" in sys.argv:
        out = bp.optimize(g3, x0, head, tail)
        x0 = out["x"]
        ts = bp.unpack_x(x0, M, D)[1]
    x0 = torch.from_numpy(np.ascontiguousarray(x0)).to(dev)
    c = torch.zeros(B, 6 * M, D, dtype=torch.float64, device=dev)
    cost = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
    g = torch.zeros(B, n, dtype=torch.float64, device=dev); s1 = torch.zeros(B, dtype=torch.int32, device=dev)
    bp._sync()
    ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(x0), pp(torch.from_numpy(head).to(dev)),
                                              pp(torch.from_numpy(tail).to(dev)), pp(cost), pp(c4), pp(g), pp(c), pp(s1)))
    co.append(c); tsl.append(ts); hl.append(head); tl.append(tail)
torch.cuda.synchronize()
coeffs = torch.cat(co); ts_a = np.ascontiguousarray(np.concatenate(tsl)); d_ts = torch.from_numpy(ts_a).to(dev)
coeffs32 = coeffs.to(torch.float32)
Ba = B * nb
order_np = npa.BatchPlanner.spatial_order(np.concatenate(hl), np.concatenate(tl))
order = torch.from_numpy(order_np).to(dev)
ns = int(np.floor(ts_a / bp.cfg.delta_t).astype(np.int64).sum())
by = ns * 32.0 + Ba * (2 * n * 4 + 20)


def bufs(dt):
    return (torch.zeros(Ba, 2, dtype=torch.float64, device=dev), torch.zeros(Ba, 6 * M, D, dtype=dt, device=dev),
            torch.zeros(Ba, M, dtype=dt, device=dev))


def launcher(mode):
    """mode: 'f64' (fp64 sampling), 'old' (round-4 body, fp64 buffers), 'seq' (sequence body, fp64 buffers), 'seq32' (fp32 buffers),
    'old32'"""
    p = npa.BatchPlanner(ctx=ctx, sample_dtype="f64" if mode == "f64" else "f32")
    if mode.startswith("old"):
        p.flags |= 8192
    else:
        p.flags |= int(os.environ.get("SEQ_FLAGS", "0"))   # (experiment libraries: -DNEO_EXPERIMENTS switches of neo_sample_seq.hpp)
    io32 = mode.endswith("32")
    out = bufs(torch.float32 if io32 else torch.float64)
    fn = ctx.lib.neo_sampled_terms_batch_f32_dev if io32 else ctx.lib.neo_sampled_terms_batch_dev
    cf = coeffs32 if io32 else coeffs

    def run():
        ctx.check(fn(ctx.h, g3.scene_id, Ba, M, D, pp(cf), pp(d_ts), pp(out[0]), pp(out[1]), pp(out[2])))
    return p, run, out


def set_order(on):
    ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(order) if on else None, 1, Ba if on else 0))


if "--only" in sys.argv:   # (under rocprofv3 --pmc: `reps` launches of one mode, nothing else)
    p, run, out = launcher(sys.argv[sys.argv.index("--only") + 1])
    p._sync(); set_order(True)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    sys.exit(0)

# ---- agreement
res_ = {}
for mode in ("f64", "old", "seq", "seq32", "old32"):
    p, run, out = launcher(mode)
    p._sync(); set_order(True)
    run(); torch.cuda.synchronize()
    res_[mode] = [o.double().cpu().numpy().copy() for o in out]
    run(); torch.cuda.synchronize()
    again = [o.double().cpu().numpy() for o in out]
    rep = all(np.array_equal(a, b_) for a, b_ in zip(res_[mode], again))
    set_order(False)
    run(); torch.cuda.synchronize()
    idx = [o.double().cpu().numpy() for o in out]
    ordi = all(np.array_equal(a, b_) for a, b_ in zip(res_[mode], idx))
    print(f"{mode:6s} reproducible {rep}  independent of the dispatch order {ordi}")
ref = res_["f64"]
for mode in ("old", "seq", "seq32", "old32"):
    errs = []
    for k in range(3):
        a, b_ = ref[k].reshape(Ba, -1), res_[mode][k].reshape(Ba, -1)
        sc = np.maximum(np.abs(a).max(axis=1), 1e-3 * np.abs(a).max())
        errs.append((np.abs(a - b_).max(axis=1) / sc).max())
    print(f"{mode:6s} vs fp64 sampling, worst trajectory: costs2 {errs[0]:.2e} grad_C {errs[1]:.2e} grad_T {errs[2]:.2e}")
for k, nm in enumerate(("costs2", "grad_C", "grad_T")):
    a, b_ = res_["old"][k].reshape(Ba, -1), res_["seq"][k].reshape(Ba, -1)
    sc = np.maximum(np.abs(a).max(axis=1), 1e-3 * np.abs(a).max())
    print(f"seq vs old {nm}: worst {(np.abs(a - b_).max(axis=1) / sc).max():.2e}, bit-equal trajectories {int((a == b_).all(axis=1).sum())} / {Ba}")
if int(os.environ.get("SEQ_FLAGS", "0")) & (1 << 18):
    print("candidates listed per trajectory: mean", res_["seq"][0][:, 1].mean(), "max", res_["seq"][0][:, 1].max())
print("violating samples: costs2 > 0 in", int((ref[0] > 0).any(axis=1).sum()), "of", Ba, "trajectories")

# ---- timing
nl, ms = ctypes.c_int64(), ctypes.c_double()


def time_mode(mode, K=20):
    p, run, out = launcher(mode)
    p._sync(); set_order(True)
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
    best1, bestk = 1e30, 1e30
    for _ in range(5):
        ctx.check(ctx.lib.neo_profile_reset(ctx.h)); ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
        ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(nl), ctypes.byref(ms)))
        best1 = min(best1, 1e3 * ms.value / max(nl.value, 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(K):
            run()
        e1.record(st)
        torch.cuda.synchronize()
        bestk = min(bestk, 1e3 * e0.elapsed_time(e1) / K)
    return best1, bestk


for rnd in range(2):
    for mode in ("old", "seq", "seq32", "old32"):
        u1, uk = time_mode(mode)
        print(f"{mode:6s} @{Ba}: one launch per event pair {u1:7.2f} us (frac_8d2 {by / (u1 * 1e-6) / 8e12:.4f}), "
              f"20 back to back {uk:7.2f} us (frac_8d2 {by / (uk * 1e-6) / 8e12:.4f})")
