#!/usr/bin/env python3
"""Every recorded reference run (tests/golden/g3_trace_*.npz) through the reference-shaped class in the fp64 mode: does the
device take the reference's number of evaluations in the last L-BFGS-B run, and how far are the finals (NEO_PLANNER_LIB
picks the library).  One line per fixture + a summary; the table behind KNOWN_PARTED of tests/test_gpu_parity.py."""
import contextlib, glob, io, os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import neo_planner_amd as npa
from neo_planner_amd import synth
import test_gpu_parity as tp
from helpers import load, rel_err
n = ex = 0
for path in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "g3_trace_*.npz"))):
    d = load(path)
    m = tp._gpu_map(d)
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    err = tp._run_entry(pl, d, m)
    last = int(d["n_runs"]) - 1
    exact = last < 0 or (pl.last_nfev == int(d[f"r{last}_nfev"]) and pl.iter_num == int(d["iter_num"]))
    n += 1; ex += exact
    xr = rel_err(pl.int_wpts, d["final_int_wpts"]) if hasattr(pl, "int_wpts") else float("nan")
    print(f"{os.path.basename(path):34s} exact {int(exact)} nfev {getattr(pl, 'last_nfev', -1):4d} ref {int(d[f'r{last}_nfev']) if last >= 0 else -1:4d} "
          f"iters {pl.iter_num} ref {int(d['iter_num'])} x_rel {xr:.2e} err {err.split(':')[0] or '-'} ref {str(d['error']).split(':')[0] or '-'}")
print(f"exact {ex} of {n}")
