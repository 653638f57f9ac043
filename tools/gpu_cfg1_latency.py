#!/usr/bin/env python3
"""BASELINE.json configs[0] on the GPU box: one replan at a time through the reference-shaped API
(plan(), M = 3, D = 2, 300 x 300 2-D map, fp64) and a full flight to the goal (30, 0); the CPU oracle
(= the reference's arithmetic, NumPy/SciPy) timed beside it on one core."""
import contextlib, io, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
from neo_planner_amd.replan import ReplanLoop
from oracle import minco_np as onp

occ = synth.occupancy_2d(3)
m = npa.ESDF(); t0 = time.perf_counter(); m.occupancy_map_cb(synth.OccupancyGridMsg(occ)); t_map = time.perf_counter() - t0
t0 = time.perf_counter(); m.occupancy_map_cb(synth.OccupancyGridMsg(occ)); t_map2 = time.perf_counter() - t0
t0 = time.perf_counter(); o = onp.GridESDF(occ, synth.RES, 300, 300, (0.0, -15.0)); t_omap = time.perf_counter() - t0
head = np.array([[0.0, 0.0], [0.0, 0.0]]); tail = np.array([[5.0, 0.3], [0.8, 0.0]])
pl = npa.MinJerkPlanner(npa.PlannerConfig())
with contextlib.redirect_stdout(io.StringIO()):
    pl.plan(m, head, tail)
    t0 = time.perf_counter()
    for _ in range(20): pl.plan(m, head, tail)
    t_gpu = (time.perf_counter() - t0) / 20
ref = onp.OraclePlanner(onp.PlannerParams())
t0 = time.perf_counter()
for _ in range(5): ref.plan(o, head, tail)
t_cpu = (time.perf_counter() - t0) / 5
print(f"ESDF build 300x300: GPU path {1e3*t_map2:.2f} ms (first call {1e3*t_map:.1f}), SciPy {1e3*t_omap:.1f} ms")
print(f"plan() M=3: GPU path {1e3*t_gpu:.2f} ms per call ({pl.last_nfev} evaluations), CPU oracle {1e3*t_cpu:.1f} ms; "
      f"final cost {pl.final_cost:.6f} vs {ref.final_cost:.6f}")
for name, planner, mp in (("GPU", npa.MinJerkPlanner(npa.PlannerConfig()), m), ("CPU oracle", onp.OraclePlanner(onp.PlannerParams()), o)):
    np.random.seed(503)
    t0 = time.perf_counter(); out = ReplanLoop(planner, mp).run(); dt = time.perf_counter() - t0
    print(f"flight to (30,0) [{name}]: {out['replans']} replans, success {out['success']}, {dt:.2f} s wall, "
          f"{1e3*dt/out['replans']:.1f} ms per replan incl. trajectory sampling, min clearance {out['min_clearance']:.2f} m")
