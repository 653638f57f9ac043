#!/usr/bin/env python3
"""GPU: every recorded reference run (tests/golden/g3_trace_*.npz) through neo_planner_amd.MinJerkPlanner with the
per-evaluation trace on (neo_optimize_trace), laid beside the reference's own recorded evaluations (r{k}_eval_f):
which fixtures end within 1e-4 of the reference's final control points, and for those that do not, the first
evaluation whose f differs (index, both values, step) -- the table tests/test_gpu_parity.py::KNOWN_PARTED quotes."""
import contextlib, ctypes, glob, io, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
import torch
import neo_planner_amd as npa
from neo_planner_amd import synth

rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))
rows = []
for path in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "g3_trace_*.npz"))):
    d = np.load(path)
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(d["occ"], float(d["res"]), d["origin"]))
    pl = npa.MinJerkPlanner(npa.PlannerConfig())
    cap = 2000
    tr = torch.zeros(1, cap, 4, dtype=torch.float64, device="cuda")
    ctx = pl.ctx
    runs = []
    orig = pl.plan_once

    def traced():
        tr.zero_()
        ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), cap))
        try:
            orig()
        finally:
            ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
            runs.append((tr[0].cpu().numpy().copy(), pl.last_nfev))
    pl.plan_once = traced
    if os.environ.get("NEO_G3_ONLY") and os.environ["NEO_G3_ONLY"] not in path:
        continue
    if int(d["np_seed"]) >= 0:
        np.random.seed(int(d["np_seed"]))
    err = ""
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            e = str(d["entry"])
            if e == "plan":
                pl.plan(m, d["head"], d["tail"])
            elif e == "batch":
                pl.batch_plan(m, d["head"], d["tail"])
            else:
                pl.read_planning_conditions(m, d["head"], d["tail"], d["init_wpts"], d["init_ts"])
                pl.plan_once()
    except Exception as ex:
        err = type(ex).__name__
    row = dict(fixture=os.path.basename(path), error=err, ref_error=str(d["error"]).split(":")[0],
               x_rel=rel(pl.int_wpts, d["final_int_wpts"]), ts_rel=rel(pl.ts, d["final_ts"]),
               iter_num=(pl.iter_num, int(d["iter_num"])), runs=(len(runs), int(d["n_runs"])))
    if "final_cost" in d.files:
        row["cost_rel"] = abs(pl.final_cost - float(d["final_cost"])) / abs(float(d["final_cost"]))
    first = None
    for k, (t, nf) in enumerate(runs):
        if f"r{k}_eval_f" not in d.files:
            break
        rf = d[f"r{k}_eval_f"]
        row.setdefault("nfev", []).append((int(nf), int(d[f"r{k}_nfev"])))
        tol = float(os.environ.get("NEO_G3_RTOL", "1e-9"))
        for j in range(min(nf, len(rf), cap)):
            if abs(t[j, 0] - rf[j]) > tol * max(abs(rf[j]), 1e-300):
                first = dict(run=k, evaluation=j, f_gpu=float(t[j, 0]), f_ref=float(rf[j]), step=float(t[j, 1]),
                             iteration=int(t[j, 3]), samples=int(t[j, 2]),
                             prev_rel_df=float(abs(t[j - 1, 0] - rf[j - 1]) / abs(rf[j - 1])) if j else None,
                             tail=[(int(t[q, 3]), float(t[q, 1]), float(t[q, 0]), float(rf[q]) if q < len(rf) else None)
                                   for q in range(max(j - 2, 0), min(nf, j + 10))])
                break
        if first:
            break
    row["first_differing_evaluation"] = first
    rows.append(row)
    print(json.dumps(row))
json.dump(rows, open(os.path.join(REPO, "gpurun_out", "g3_report.json"), "w"), indent=1)
