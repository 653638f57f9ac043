#!/usr/bin/env python3
"""all-fp32 mode (NEO_FLAG_F32_SOLVE) against the default modes on a GPU box: per-evaluation agreement and optimiser
statistics"""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import neo_planner_amd as npa
from neo_planner_amd import synth
dist = synth.esdf_3d(2, n=100, res=0.3)
g3 = npa.ESDF3D(dist, 0.3, synth.DOMAIN_ORIGIN, store="f32", layout="yz4")
for M in (2, 3, 5, 8, 11, 21, 31, 41):
    B = 256
    head, tail, wp, ts = synth.replan_requests(2, B, M - 1, D=3, **synth.VOLUME)
    o = {}
    for dt in ("f64", "f32", "f32x"):
        bp = npa.BatchPlanner(sample_dtype=dt)
        o[dt] = bp.cost_grad(g3, bp.pack_x(wp, ts), head, tail, want_coeffs=True)
    def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
    def relrow(a, b): return float(np.median(np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)))
    print(f"M={M:2d} f32x vs f64: coeff {rel(o['f32x']['coeffs'], o['f64']['coeffs']):.2e} cost {rel(o['f32x']['cost'], o['f64']['cost']):.2e} "
          f"grad(max) {rel(o['f32x']['grad'], o['f64']['grad']):.2e} grad(median row) {relrow(o['f32x']['grad'], o['f64']['grad']):.2e} | "
          f"f32 vs f64: cost {rel(o['f32']['cost'], o['f64']['cost']):.2e} grad(median row) {relrow(o['f32']['grad'], o['f64']['grad']):.2e}")
M, B = 21, 2048
head, tail, wp, ts = synth.replan_requests(3, B, M - 1, D=3, **synth.VOLUME)
w = None
for dt in ("f64", "f32", "f32x"):
    bp = npa.BatchPlanner(sample_dtype=dt, waves_per_simd=2 if dt != "f64" else None)
    r = bp.optimize(g3, bp.pack_x(wp, ts), head, tail)
    st, cnt = np.unique(r["status"], return_counts=True)
    print(dt, "mean nfev %.1f  mean nit %.1f  median final cost %.5f  status" % (r["nfev"].mean(), r["nit"].mean(), np.median(r["final_cost"])), dict(zip(st.tolist(), cnt.tolist())))
