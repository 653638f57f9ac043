"""Latency of the reference-shaped callbacks get_cost(x) / get_grad(x) (expert_planner.py:539-585) for one trajectory of
M = 3 on the 2-D map: one neo_cost_grad_batch call each, through the pinned small-call path."""
import os, sys, time, contextlib, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
occ = synth.occupancy_2d(3)
m = npa.ESDF(); m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
head = np.array([[0.0, 0.0], [0.0, 0.0]]); tail = np.array([[5.0, 0.3], [0.8, 0.0]])
pl = npa.MinJerkPlanner(npa.PlannerConfig())
with contextlib.redirect_stdout(io.StringIO()):
    pl.plan(m, head, tail)
x = np.concatenate([pl.int_wpts.reshape(-1), pl.tau])
for f in (pl.get_cost, pl.get_grad):
    f(x); t0 = time.perf_counter()
    for _ in range(200): f(x)
    print(f.__name__, f"{1e6 * (time.perf_counter() - t0) / 200:.0f} us per call")
