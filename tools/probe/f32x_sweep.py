#!/usr/bin/env python3
"""all-fp32 mode against the mixed-precision mode over problem sizes (GPU box): statuses, effort, final costs"""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import neo_planner_amd as npa
from neo_planner_amd import synth
dist = synth.esdf_3d(1, n=150, res=0.2, canopy=40)
for store, layout in (("f32", "yz4"), ("f16", "linear")):
    g3 = npa.ESDF3D(dist, 0.2, synth.DOMAIN_ORIGIN, store=store, layout=layout)
    for M in (2, 3, 5, 8, 13, 21, 25, 31, 41, 50, 64):
        B = 512
        head, tail, wp, ts = synth.replan_requests(M, B, M - 1, D=3, **synth.VOLUME)
        r = {}
        for dt in ("f32", "f32x"):
            bp = npa.BatchPlanner(sample_dtype=dt, waves_per_simd=2)
            r[dt] = bp.optimize(g3, bp.pack_x(wp, ts), head, tail)
        a, b = r["f32"], r["f32x"]
        ok = (a["status"] <= 1) & (b["status"] <= 1)
        fin = np.isfinite(b["x"]).all()
        print(f"{store} {layout} M={M:2d}: ok f32 {(a['status']<=1).mean():.3f} f32x {(b['status']<=1).mean():.3f}  nfev {a['nfev'].mean():.1f} / {b['nfev'].mean():.1f}  "
              f"median cost {np.median(a['final_cost'][ok]):.4f} / {np.median(b['final_cost'][ok]):.4f}  statuses f32x {dict(zip(*[x.tolist() for x in np.unique(b['status'], return_counts=True)]))}  finite {fin}")
