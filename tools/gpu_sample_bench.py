#!/usr/bin/env python3
"""Times variants of the stand-alone ESDF sample kernel (tools/probe/sample_variants.hip) against the
production `sample_kernel` on the cfg2 batch, at the initial guess and at the optimised trajectories, and
checks every variant's outputs against the production kernel's.

    python tools/gpu_sample_bench.py            # on a GPU box; builds the probe library if needed
"""
import ctypes
import os
import subprocess
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np

PROBE_SRC = os.path.join(REPO, "tools", "probe", "sample_variants.hip")
PROBE_LIB = os.path.join(REPO, "tools", "probe", "_build", "libsample_variants.so")


def build_probe():
    deps = [PROBE_SRC, os.path.join(REPO, "neo-planner_amd", "csrc", "neo_device.hpp")]
    if os.path.exists(PROBE_LIB) and all(os.path.getmtime(PROBE_LIB) > os.path.getmtime(d) for d in deps):
        return
    os.makedirs(os.path.dirname(PROBE_LIB), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-w",
                           "-I", os.path.join(REPO, "include"), PROBE_SRC, "-o", PROBE_LIB])


def main():
    build_probe()
    if "--build-only" in sys.argv:
        return
    import torch
    import neo_planner_amd as npa
    from neo_planner_amd import synth, _lib

    dev = torch.device("cuda:0")
    grid = 300
    res = 30.0 / grid
    dist = synth.esdf_3d(0, n=grid, res=res).astype(np.float32)
    B, M, D = 4096, 21, 3
    head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=3)
    g3 = npa.ESDF3D(dist, res, synth.DOMAIN_ORIGIN, store="f32")
    bp = npa.BatchPlanner(sample_dtype="f32")
    x0 = bp.pack_x(wp, ts)
    e0 = bp.cost_grad(g3, x0, head, tail, want_coeffs=True)
    opt = bp.optimize(g3, x0, head, tail)
    _, ts1 = bp.unpack_x(opt["x"], M, D)
    e1 = bp.cost_grad(g3, opt["x"], head, tail, want_coeffs=True)
    cases = {"initial guess": (e0["coeffs"], ts), "optimised": (e1["coeffs"], ts1)}

    P = ctypes.CDLL(PROBE_LIB)
    P.probe_run.restype = ctypes.c_double
    P.probe_name.restype = ctypes.c_char_p
    P.probe_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_double] + \
        [ctypes.c_void_p] * 6 + [ctypes.c_int]
    cfg = bp.cfg
    prm = np.array([cfg.v_max, cfg.T_min, cfg.T_max, cfg.safe_dis, cfg.delta_t] + list(cfg.weights), dtype=np.float64)
    org = np.asarray(synth.DOMAIN_ORIGIN, dtype=np.float64)
    field = torch.from_numpy(dist).to(dev)
    field = torch.cat([field.reshape(-1), torch.zeros(64, device=dev)])
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    ctx = bp.ctx
    for name, (co, tt) in cases.items():
        n_samples = int(np.floor(tt / cfg.delta_t).astype(np.int64).sum())
        by = n_samples * 32.0 + B * (2 * 6 * M * D * 8 + 2 * M * 8 + 16)
        d_co = torch.from_numpy(np.ascontiguousarray(co)).to(dev)
        d_ts = torch.from_numpy(np.ascontiguousarray(tt)).to(dev)
        c2 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        gC = torch.zeros_like(d_co)
        gT = torch.zeros(B, M, dtype=torch.float64, device=dev)
        # production kernel
        bp._sync()
        run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(d_co), pp(d_ts),
                                                                    pp(c2), pp(gC), pp(gT)))
        def production():
            for _ in range(5):
                run()
            ctx.check(ctx.lib.neo_ctx_synchronize(ctx.h))
            ctx.check(ctx.lib.neo_profile_reset(ctx.h))
            ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
            for _ in range(50):
                run()
            ctx.check(ctx.lib.neo_ctx_synchronize(ctx.h))
            ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
            l2 = ctypes.c_int64(); m2 = ctypes.c_double()
            ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(l2), ctypes.byref(m2)))
            return 1e3 * m2.value / max(l2.value, 1)
        us = production()
        ref = [c2.cpu().numpy().copy(), gC.cpu().numpy().copy(), gT.cpu().numpy().copy()]
        print(f"== {name}: {n_samples} samples, {by / 1e6:.1f} MB algorithmic")
        nsp = np.floor(tt / cfg.delta_t).astype(np.int64)
        Lu = 64 // M
        r_uniform = np.ceil(nsp / Lu).max(axis=1)
        r_ideal = np.ceil(nsp.sum(axis=1) / 64)
        r_adapt = np.zeros(B)
        for b_ in range(B):
            R = int(r_ideal[b_])
            while np.ceil(nsp[b_] / R).sum() > 64:
                R += 1
            r_adapt[b_] = R
        print(f"   sample rounds per trajectory: uniform L={Lu}: {r_uniform.mean():.2f}, lanes ~ samples: {r_adapt.mean():.2f}, "
              f"ideal {r_ideal.mean():.2f}; samples/piece min {nsp.min()} mean {nsp.mean():.1f} max {nsp.max()}")
        print(f"   production sample_kernel      {us:7.2f} us  {by / us / 1e3 / 8000 * 100:5.1f} % of 8 TB/s")
        for v in range(P.probe_variants()):
            c2.zero_(); gC.zero_(); gT.zero_()
            us = P.probe_run(v, B, M, prm.ctypes.data, pp(field), grid, grid, grid, res, org.ctypes.data,
                             pp(d_co), pp(d_ts), pp(c2), pp(gC), pp(gT), 50)
            torch.cuda.synchronize()
            out = [c2.cpu().numpy(), gC.cpu().numpy(), gT.cpu().numpy()]
            err = [float(np.abs(o - r_).max() / max(np.abs(r_).max(), 1e-30)) for o, r_ in zip(out, ref)]
            print(f"   v{v} {P.probe_name(v).decode():<22} {us:7.2f} us  {by / us / 1e3 / 8000 * 100:5.1f} %   "
                  f"max err / max |ref|: costs {err[0]:.1e} gC {err[1]:.1e} gT {err[2]:.1e}")
            if v < 4:
                for dbg, what in ((1, "no sample loop"), (2, "at most 2 rounds"), (4, "one event pair per launch"), (5, "same, no sample loop")):
                    us = P.probe_run(v | (dbg << 8), B, M, prm.ctypes.data, pp(field), grid, grid, grid, res, org.ctypes.data,
                                     pp(d_co), pp(d_ts), pp(c2), pp(gC), pp(gT), 50)
                    print(f"        ({what}: {us:7.2f} us)")
        print(f"   production sample_kernel again {production():6.2f} us")
        ctx.set_params(flags=1)
        print(f"        (no sample loop: {production():6.2f} us)")
        ctx.set_params(flags=0)


if __name__ == "__main__":
    main()
