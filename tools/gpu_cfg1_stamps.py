#!/usr/bin/env python3
"""Phase times of ONE plan() of the reference's shape (M = 3, D = 2, 300 x 300 nearest-cell map, fp64) -- a wavefront
alone on the chip.  Needs a library built with NEO_BUILD_DEFS=-DNEO_STAMPS (NEO_PLANNER_LIB=...)."""
import ctypes, os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth

occ = synth.occupancy_2d(3)
m = npa.ESDF(); m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
from neo_planner_amd import _lib
ctx = _lib.default_context()
head = np.array([[[0.0, 0.0], [0.0, 0.0], [0.0, 0.0]]]); tail = np.array([[[5.0, 0.3], [0.8, 0.0], [0.0, 0.0]]])
B = int(os.environ.get("NEO_B", "1"))
head = np.repeat(head, B, 0); tail = np.repeat(tail, B, 0)
bp = npa.BatchPlanner(ctx=ctx, sample_dtype=os.environ.get("NEO_DTYPE", "f64"))
pl = npa.MinJerkPlanner(npa.PlannerConfig())
wp, ts = pl.generate_init_variables(head[0], tail[0])
wp = np.repeat(np.asarray(wp)[None], B, 0); ts = np.repeat(np.asarray(ts)[None], B, 0)
x0 = bp.pack_x(wp, ts)
dev = torch.device("cuda", 0)
cnt = torch.zeros(B, 8, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.neo_optimize_sample_counter(ctx.h, ctypes.c_void_p(cnt.data_ptr())))
for _ in range(3):
    out = bp.optimize(m, x0, head, tail)
torch.cuda.synchronize()
c = cnt.cpu().numpy().astype(np.float64); tick = 0.01
ev = c[:, 1]
print(f"B = {B}: evaluations {ev.mean():.1f}, status {out['status'][:4]}, samples/eval {c[:, 0].sum() / ev.sum():.1f}")
e = ev.sum()
f, s_, b_ = c[:, 2].sum() * tick / e, c[:, 3].sum() * tick / e, c[:, 4].sum() * tick / e
t = c[:, 5].sum() * tick / e; tl = c[:, 7].sum() * tick / e
print(f"per evaluation: total {t:.2f} us = forward {f:.2f} + sample {s_:.2f} + backward {b_:.2f} + optimiser {t - f - s_ - b_:.2f} (two-loop {tl:.2f}); run {c[:, 5].mean() * tick:.0f} us")
