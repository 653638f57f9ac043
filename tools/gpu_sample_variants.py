#!/usr/bin/env python3
"""Times the variants of the workgroup-per-trajectory ESDF-lookup kernel (tools/probe/neo_sample_wg.hpp; a library built with
-DNEO_SAMPLE_EXPERIMENTS) against the one-wavefront sample_kernel on bench.py's cfg2 workload (300^3 fp32 field, yz-quad
layout, requests filling the volume), one launch of 4096 trajectories and one of 65 536, and checks every variant's
outputs against the one-wavefront kernel's.

    NEO_BUILD_DEFS=-DNEO_SAMPLE_EXPERIMENTS NEO_BUILD_OUT=_build/libneo_sample_exp.so python -m neo_planner_amd.build      (here)
    NEO_PLANNER_LIB=_build/libneo_sample_exp.so python tools/gpu_sample_variants.py                                (GPU box)
    python tools/gpu_sample_variants.py [variant ...]                            (GPU box)
"""
import ctypes
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np


def main():
    import torch
    import neo_planner_amd as npa
    from neo_planner_amd import synth, _lib
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 412, 414, 423, 443, 314, 316, 323, 325, 343]
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    occ = synth.occupancy_3d(0, n=300, res=0.1, canopy=80)
    g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), 0.1, synth.DOMAIN_ORIGIN, store="f32", layout="yz4", ctx=ctx)
    M, D = 21, 3
    n = D * (M - 1) + M
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32")
    bp._sync()
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    out = {}
    for nb in (1, 16):
        B = 4096 * nb
        sets = [synth.replan_requests(1000 * r, 4096, M - 1, D=3, **synth.VOLUME) for r in range(nb)]
        head, tail, wp, ts = (np.concatenate([s_[k] for s_ in sets]) for k in range(4))
        x0 = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
        d_h, d_t = torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev)
        coeffs = torch.zeros(B, 6 * M, D, dtype=torch.float64, device=dev)
        cost1 = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
        grad1 = torch.zeros(B, n, dtype=torch.float64, device=dev); st1 = torch.zeros(B, dtype=torch.int32, device=dev)
        ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(x0), pp(d_h), pp(d_t), pp(cost1), pp(c4),
                                                  pp(grad1), pp(coeffs), pp(st1)))
        d_ts = torch.from_numpy(np.ascontiguousarray(ts)).to(dev)
        ns = int(np.floor(ts / bp.cfg.delta_t).astype(np.int64).sum())
        by = ns * 32.0 + B * (2 * n * 4 + 20)
        ref = None
        for v in variants:
            os.environ["NEO_SAMPLE_VARIANT"] = str(v)       # 0: the product's sample_kernel, 1: contiguous chunks (tools/probe/neo_sample_chunk.hpp), else neo_sample_wg.hpp
            c2 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
            gC = torch.zeros_like(coeffs); gT = torch.zeros(B, M, dtype=torch.float64, device=dev)
            run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(coeffs), pp(d_ts),
                                                                        pp(c2), pp(gC), pp(gT)))
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            ctx.check(ctx.lib.neo_profile_reset(ctx.h)); ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
            for _ in range(40 if nb == 1 else 15):
                run()
            torch.cuda.synchronize()
            ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
            l2 = ctypes.c_int64(); m2 = ctypes.c_double()
            ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(l2), ctypes.byref(m2)))
            us = 1e3 * m2.value / max(l2.value, 1)
            res = (c2.cpu().numpy(), gC.cpu().numpy(), gT.cpu().numpy())
            if ref is None:
                ref = res
            err = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) for a, b in zip(res, ref)]
            # run-to-run reproducibility
            run(); torch.cuda.synchronize()
            same = bool(np.array_equal(gC.cpu().numpy(), res[1]) and np.array_equal(c2.cpu().numpy(), res[0]))
            out[f"B{B}_v{v}"] = dict(us=us, frac_8d2=by / (us * 1e-6) / 8e12, max_rel_diff_vs_first=err, reproducible=same)
            print(f"B {B:6d} variant {v:4d}: {us:8.2f} us  frac_8d2 {by / (us * 1e-6) / 8e12:.3f}  diff vs first {err}  repro {same}",
                  flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
