#!/usr/bin/env python3
"""ESDF-lookup kernel (sample_kernel) on bench.py's cfg2 / cfg5 request batches: field layouts x dispatch orders.

    python tools/gpu_esdf_locality.py [--config cfg2|cfg5] [--sets 40] [--json out.json]

For every layout (yz4, brick, ...) and dispatch order (index, XCD-aware spatial orders of BatchPlanner.spatial_order) it
times the 4096-trajectory launch and the launch over all request batches of a bench step (HIP events on the kernel's
stream, the bench's own protocol) and checks that every variant returns the same bits.  Under rocprofv3 --pmc the kernel
launches are told apart by their order in the trace (variants run in the printed order).
"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg5"])
    ap.add_argument("--sets", type=int, default=None)
    ap.add_argument("--layouts", default="yz4,brick")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only-order", default=None, help="run a single order (for counter passes)")
    a = ap.parse_args()
    import torch
    import neo_planner_amd as npa
    from neo_planner_amd import synth, _lib

    grid, wpn, store, n_sets = (300, 20, "f32", 40) if a.config == "cfg2" else (600, 40, "f16", 4)
    if a.sets:
        n_sets = a.sets
    M, D, B = wpn + 1, 3, 4096
    n = D * (M - 1) + M
    esz = 4 if store == "f32" else 2
    res = 30.0 / grid
    dev = torch.device("cuda", 0)
    ctx = npa.Context(0)
    occ = synth.occupancy_3d(0, n=grid, res=res, canopy=80)
    d_occ = torch.from_numpy(occ).to(dev)
    sets = [synth.replan_requests(1000 * r, B, M - 1, D=D, **synth.VOLUME) for r in range(n_sets)]
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32")
    bp._sync()
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    Ba = B * n_sets
    head_a = np.concatenate([s[0] for s in sets]); tail_a = np.concatenate([s[1] for s in sets])
    ts_a = np.ascontiguousarray(np.concatenate([s[3] for s in sets]))
    ns_piece = np.floor(ts_a / bp.cfg.delta_t).astype(np.int64)
    samples = {B: int(ns_piece[:B].sum()), Ba: int(ns_piece.sum())}
    orders = {"index": None, "spatial 1 m": lambda h, t: npa.BatchPlanner.spatial_order(h, t, cell=1.0),
              "spatial 0.5 m": lambda h, t: npa.BatchPlanner.spatial_order(h, t, cell=0.5),
              "spatial 2 m": lambda h, t: npa.BatchPlanner.spatial_order(h, t, cell=2.0),
              "sorted, no XCD deal": lambda h, t: npa.BatchPlanner.spatial_order(h, t, cell=1.0, xcds=1),
              "spatial, runs of 64": lambda h, t: npa.BatchPlanner.spatial_order(h, t, chunk=64),
              "spatial, runs of 256": lambda h, t: npa.BatchPlanner.spatial_order(h, t, chunk=256),
              "spatial, runs of 512": lambda h, t: npa.BatchPlanner.spatial_order(h, t, chunk=512),
              "spatial, runs of 2048": lambda h, t: npa.BatchPlanner.spatial_order(h, t, chunk=2048)}
    if a.only_order:
        orders = {k: v for k, v in orders.items() if k == a.only_order}
    results = []
    ref_out = {}
    coeffs_a = None
    for layout in a.layouts.split(","):
        g3 = npa.ESDF3D.from_occupancy(d_occ, res, synth.DOMAIN_ORIGIN, store=store, layout=layout, ctx=ctx)
        if coeffs_a is None:
            coeffs_a = torch.zeros(Ba, 6 * M, D, dtype=torch.float64, device=dev)
            cost1 = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
            grad1 = torch.zeros(B, n, dtype=torch.float64, device=dev); st1 = torch.zeros(B, dtype=torch.int32, device=dev)
            for r_, s_ in enumerate(sets):
                x0 = torch.from_numpy(bp.pack_x(s_[2], s_[3])).to(dev)
                h_ = torch.from_numpy(s_[0]).to(dev); t_ = torch.from_numpy(s_[1]).to(dev)
                ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, pp(x0), pp(h_), pp(t_), pp(cost1), pp(c4),
                                                          pp(grad1), pp(coeffs_a[r_ * B:(r_ + 1) * B]), pp(st1)))
            torch.cuda.synchronize()
        d_ts = torch.from_numpy(ts_a).to(dev)
        for nb in (B, Ba):
            c2 = torch.zeros(nb, 2, dtype=torch.float64, device=dev)
            gC = torch.zeros(nb, 6 * M, D, dtype=torch.float64, device=dev)
            gT = torch.zeros(nb, M, dtype=torch.float64, device=dev)
            run = lambda: ctx.check(ctx.lib.neo_sampled_terms_batch_dev(ctx.h, g3.scene_id, nb, M, D, pp(coeffs_a), pp(d_ts),
                                                                        pp(c2), pp(gC), pp(gT)))
            for oname, fn in orders.items():
                od = None
                if fn is not None:
                    od = torch.from_numpy(fn(head_a[:nb], tail_a[:nb])).to(dev)
                ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(od) if od is not None else None, 1, nb))
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                ctx.check(ctx.lib.neo_profile_reset(ctx.h))
                ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
                for _ in range(a.reps if nb == B else max(a.reps // 3, 5)):
                    run()
                torch.cuda.synchronize()
                ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
                l2 = ctypes.c_int64(); m2 = ctypes.c_double()
                ctx.check(ctx.lib.neo_profile_read(ctx.h, _lib.NEO_KERNEL_ESDF_SAMPLE, ctypes.byref(l2), ctypes.byref(m2)))
                us = 1e3 * m2.value / max(l2.value, 1)
                by = samples[nb] * 8.0 * esz + nb * (2 * n * 4 + 20)
                out = (c2.cpu().numpy().copy(), gC.cpu().numpy().copy(), gT.cpu().numpy().copy())
                key = nb
                same = True
                if key in ref_out:
                    same = all(np.array_equal(x, y) for x, y in zip(out, ref_out[key]))
                else:
                    ref_out[key] = out
                row = dict(config=a.config, layout=layout, order=oname, trajectories=nb, kernel_us=us,
                           frac_8d2=by / (us * 1e-6) / 1e9 / 8000.0, same_bits_as_first_variant=bool(same))
                results.append(row)
                print(f"{a.config} {layout:6s} {oname:22s} B={nb:6d}  {us:9.2f} us  frac_8d2 {row['frac_8d2']:.3f}  same bits {same}", flush=True)
        ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, None, 0, 0))
        ctx.check(ctx.lib.neo_esdf_drop(ctx.h, g3.scene_id))
    if a.json:
        json.dump(results, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
