#!/usr/bin/env python3
"""
The three arithmetic modes of the HIP path against the REFERENCE-GENERATED fixtures.

Everything here runs the product on the reference's own 2-D nearest-cell map (map_server/esdf.py:53-82) through the
reference-shaped class `MinJerkPlanner` (expert_planner.py:205-237, 539-585) in the modes

    f64    fp64 everywhere (the parity mode)
    f32    fp32 sampled terms, fp64 solve / adjoint / optimiser
    f32x   everything fp32 (NEO_FLAG_F32_SOLVE) -- the arithmetic bench.py's `value` is measured in

and compares with what the real reference produced on the same inputs (tests/golden, tools/gen_golden.py):

    G1  per evaluation: cost, four cost terms, gradient, coefficients            (g1_eval_s*.npz)
    G3  recorded optimiser runs: exceptions, evaluation counts, finals           (g3_trace_*.npz)
    G6  256 M = 21 plan_once runs: share of finals within north_star's 1e-4 of the reference's, next to the share on
        which the reference under another BLAS kernel set stays within 1e-4 of ITSELF; exit statuses on both sides
        (g6_reference_vs_itself.npz)

Used by tests/test_gpu_reference_fixtures.py (thresholds), by bench.py (`parity.vs_reference_fixtures`) and stand-alone:

    python tools/ref_fixture_parity.py [out.json]      # on the GPU box; the summary kept under profiles/
"""
import contextlib
import glob
import io
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (REPO, os.path.join(REPO, "neo-planner_amd")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
GOLDEN = os.path.join(REPO, "tests", "golden")
MODES = ("f64", "f32", "f32x")
G6_BASE = "blas_threads_1"
G6_OTHER = ("blas_coretype_haswell", "blas_coretype_sandybridge")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _gpu_map(npa, synth, occ, res, origin):
    m = npa.ESDF()
    m.occupancy_map_cb(synth.OccupancyGridMsg(occ, float(res), origin))
    return m


# ------------------------------------------------------------------------------------------------ G1
G1_TOL = {"f64": 1e-10, "f32": 2e-5, "f32x": 4e-5}


def g1_report(modes=MODES, jump_fn=None):
    """the 8 fixtures x M in {3, 21, 41}: relative deviations of one evaluation from the reference's, per mode the maximum,
    the median and every case beyond the mode's tolerance -- with the reference objective's own jump there when the caller
    supplies `jump_fn(d, M)` (tests/helpers.py:reference_jump evaluates it with the pinned oracle; this module, which
    bench.py's GPU process imports, never touches oracle/)"""
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    out = {m: dict(tolerance=G1_TOL[m], cost=[], costs=[], grad=[], coeffs=[], beyond=[]) for m in modes}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "g1_eval_s*.npz"))):
        d = np.load(path)
        mp = _gpu_map(npa, synth, d["occ"], d["res"], d["origin"])
        for mode in modes:
            bp = npa.BatchPlanner(sample_dtype=mode)
            for M in (3, 21, 41):
                t = f"M{M}_"
                r = bp.cost_grad(mp, d[t + "x"][None], d[t + "head"][None], d[t + "tail"][None], want_coeffs=True)
                o = out[mode]
                e = dict(cost=abs(r["cost"][0] - float(d[t + "cost"])) / abs(float(d[t + "cost"])),
                         costs=rel_err(r["costs"][0], d[t + "costs"]), grad=rel_err(r["grad"][0], d[t + "grad"]),
                         coeffs=rel_err(r["coeffs"][0], d[t + "coeffs"]))
                for k, v in e.items():
                    o[k].append(float(v))
                if max(e["cost"], e["costs"], e["grad"]) > G1_TOL[mode]:
                    o["beyond"].append(dict(fixture=os.path.basename(path), M=M, cost=float(e["cost"]), costs=float(e["costs"]),
                                            grad=float(e["grad"])))
                    if jump_fn is not None:
                        cj, gj = jump_fn(d, M)
                        o["beyond"][-1]["reference_jump_under_4e_6_noise"] = cj
                        o["beyond"][-1]["reference_gradient_jump_under_4e_6_noise"] = gj
    for o in out.values():
        o["n"] = len(o["cost"])
        for k in ("cost", "costs", "grad", "coeffs"):
            v = np.array(o.pop(k))
            o[k + "_max"] = float(v.max())
            o[k + "_median"] = float(np.median(v))
        o["within_tolerance"] = o["n"] - len(o["beyond"])
    return out


# ------------------------------------------------------------------------------------------------ G3
def run_entry(pl, d, m):
    entry = str(d["entry"])
    if int(d["np_seed"]) >= 0:
        np.random.seed(int(d["np_seed"]))
    err = ""
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            if entry == "plan":
                pl.plan(m, d["head"], d["tail"])
            elif entry == "batch":
                pl.batch_plan(m, d["head"], d["tail"])
            else:
                pl.read_planning_conditions(m, d["head"], d["tail"], d["init_wpts"], d["init_ts"])
                pl.plan_once()
    except Exception as ex:
        err = f"{type(ex).__name__}:{ex}"
    return err


def g3_report(modes=MODES):
    """every recorded reference run (plan, replans, batch_plan, seeded retries, M = 21 / 41 plan_once) in every mode"""
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    rows = {m: [] for m in modes}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "g3_trace_*.npz"))):
        d = np.load(path)
        mp = _gpu_map(npa, synth, d["occ"], d["res"], d["origin"])
        last = int(d["n_runs"]) - 1
        for mode in modes:
            pl = npa.MinJerkPlanner(npa.PlannerConfig(), sample_dtype=mode)
            err = run_entry(pl, d, mp)
            row = dict(fixture=os.path.basename(path), entry=str(d["entry"]), error=err.split(":")[0],
                       ref_error=str(d["error"]).split(":")[0], iter_num=int(pl.iter_num), ref_iter_num=int(d["iter_num"]),
                       runs=int(pl.opt_running_times), ref_runs=int(d["opt_running_times"]),
                       nfev_last=int(getattr(pl, "last_nfev", -1)), ref_nfev_last=int(d[f"r{last}_nfev"]) if last >= 0 else -1)
            if hasattr(pl, "int_wpts") and np.shape(pl.int_wpts) == np.shape(d["final_int_wpts"]):
                row["x_rel"] = rel_err(pl.int_wpts, d["final_int_wpts"])
                row["ts_rel"] = rel_err(pl.ts, d["final_ts"])
            if "final_cost" in d.files and hasattr(pl, "final_cost"):
                row["cost_rel"] = abs(pl.final_cost - float(d["final_cost"])) / abs(float(d["final_cost"]))
            rows[mode].append(row)
    return rows


def g3_summary(rows):
    out = {}
    for mode, rs in rows.items():
        x = np.array([r.get("x_rel", np.nan) for r in rs])
        c = np.array([r.get("cost_rel", np.nan) for r in rs])
        out[mode] = dict(
            n=len(rs), same_exception=int(sum(r["error"] == r["ref_error"] for r in rs)),
            same_run_count=int(sum(r["runs"] == r["ref_runs"] for r in rs)),
            same_nfev_last=int(sum(r["nfev_last"] == r["ref_nfev_last"] for r in rs)),
            finals_within_1e_4=int(np.sum(x <= 1e-4)), finals_within_1e_2=int(np.sum(x <= 1e-2)),
            cost_within_1e_4=int(np.sum(c <= 1e-4)), cost_within_1e_2=int(np.sum(c <= 1e-2)),
            x_rel_median=float(np.nanmedian(x)), cost_rel_median=float(np.nanmedian(c)))
    return out


# ------------------------------------------------------------------------------------------------ G6
NQ = 40


def _g6_finals(d, env):
    n = int(d["n_requests"])
    x = np.stack([d[f"{env}__q{k}_x"] for k in range(n)])
    nfev = np.array([int(d[f"{env}__q{k}_nfev"]) for k in range(n)])
    err = [str(d[f"{env}__q{k}_error"]) for k in range(n)]
    fun = np.array([float(d[f"{env}__q{k}_fun"]) for k in range(n)])
    msg = [str(d[f"{env}__q{k}_message"]) if f"{env}__q{k}_message" in d.files else "" for k in range(n)]
    return x, nfev, err, fun, msg


def _dx(x, ref):
    return np.abs(x[:, :NQ] - ref[:, :NQ]).max(axis=1) / np.abs(ref[:, :NQ]).max(axis=1)


def scipy_message_code(msg):
    """SciPy's L-BFGS-B exit message -> the NEO_TRAJ_* code of the same exit (include/neo_planner.h)"""
    m = msg.upper()
    if "PROJECTED GRADIENT" in m:
        return 0
    if "REDUCTION OF F" in m:
        return 1
    if "ABNORMAL" in m:
        return 2
    if "LIMIT" in m or "MAXIMUM" in m:
        return 3
    return -1


EXIT_NAMES = ["CONVERGED_GRAD", "CONVERGED_F", "ABNORMAL", "MAXITER", "NUMERIC_RANGE", "NONFINITE", "BAD_SCENE"]


def _hist_codes(codes):
    h = {}
    for c in codes:
        name = EXIT_NAMES[c] if 0 <= c < len(EXIT_NAMES) else "none"
        h[name] = h.get(name, 0) + 1
    return h


def _hist(strings):
    h = {}
    for s in strings:
        s = s.split(":")[0] or "none"
        h[s] = h.get(s, 0) + 1
    return h


def g6_reference_side(d):
    """the reference against itself under another BLAS kernel set, and its own exits"""
    xb, nb, eb, fb, mb = _g6_finals(d, G6_BASE)
    ok = nb > 0
    out = dict(n=int(ok.sum()), exceptions=_hist([e for e, k in zip(eb, ok) if k]),
               exits=_hist_codes([scipy_message_code(m) for m, k in zip(mb, ok) if k]) if any(mb) else None,
               mean_nfev=float(nb[ok].mean()))
    for env in G6_OTHER:
        x, nf, er, fu, _ = _g6_finals(d, env)
        sel = ok & (nf > 0)
        dx = _dx(x[sel], xb[sel])
        rc = np.abs(fu[sel] - fb[sel]) / np.maximum(np.abs(fb[sel]), 1e-300)
        out[env] = dict(n=int(sel.sum()), finals_within_1e_4=float((dx <= 1e-4).mean()), same_nfev=float((nf[sel] == nb[sel]).mean()),
                        cost_within_1e_4=float((rc <= 1e-4).mean()), x_rel_median=float(np.median(dx)),
                        same_exception=float(np.mean([a.split(":")[0] == b.split(":")[0]
                                                      for a, b, k in zip(er, eb, sel) if k])))
    out["self_agreement_min"] = min(out[e]["finals_within_1e_4"] for e in G6_OTHER)
    return out


def g6_device_side(d, modes=MODES, limit=None):
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    xb, nb, eb, fb, mb = _g6_finals(d, G6_BASE)
    ok = nb > 0
    n = int(d["n_requests"]) if limit is None else min(int(d["n_requests"]), limit)
    maps = {}
    out = {}
    for mode in modes:
        finals = np.full_like(xb, np.nan)
        nfev = np.zeros(len(xb), dtype=int)
        cost = np.full(len(xb), np.nan)
        errs, codes = [""] * len(xb), [-1] * len(xb)
        for k in range(n):
            if not ok[k]:
                continue
            ms = int(d[f"q{k}_map_seed"])
            if ms not in maps:
                maps[ms] = _gpu_map(npa, synth, d[f"occ{ms}"], d["res"], d["origin"])
            pl = npa.MinJerkPlanner(npa.PlannerConfig(), sample_dtype=mode)
            pl.read_planning_conditions(maps[ms], d[f"q{k}_head"], d[f"q{k}_tail"], d[f"q{k}_init_wpts"], d[f"q{k}_init_ts"])
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    pl.plan_once()
            except Exception as ex:
                errs[k] = f"{type(ex).__name__}:{ex}"
            codes[k] = int(getattr(pl, "last_status", -1))
            if hasattr(pl, "tau") and not errs[k].startswith("OverflowError"):
                finals[k] = np.concatenate([np.reshape(pl.int_wpts, -1), pl.tau])
                nfev[k] = pl.last_nfev
                cost[k] = float(np.dot(pl.costs_at_x, pl.weights))
        sel = ok & np.isfinite(finals[:, 0])
        sel[n:] = False
        dx = _dx(finals[sel], xb[sel])
        rc = np.abs(cost[sel] - fb[sel]) / np.maximum(np.abs(fb[sel]), 1e-300)
        run = ok.copy()
        run[n:] = False
        out[mode] = dict(n=int(sel.sum()), finals_within_1e_4=float((dx <= 1e-4).mean()),
                         finals_within_1e_2=float((dx <= 1e-2).mean()), same_nfev=float((nfev[sel] == nb[sel]).mean()),
                         cost_within_1e_4=float((rc <= 1e-4).mean()), cost_within_1e_2=float((rc <= 1e-2).mean()),
                         x_rel_median=float(np.median(dx)), cost_rel_median=float(np.median(rc)),
                         mean_nfev=float(nfev[sel].mean()),
                         median_final_cost=float(np.median(cost[sel])), reference_median_final_cost=float(np.median(fb[sel])),
                         # run by run: final cost over the reference's (1 = the same solution; the optimiser is a descent
                         # method on a non-convex objective: runs that part from the reference's path end in another local
                         # minimum, better or worse)
                         cost_ratio_quantiles={q: float(np.quantile(cost[sel] / fb[sel], q / 100.0)) for q in (5, 25, 50, 75, 95)},
                         cost_ratio_log_mean=float(np.mean(np.log(cost[sel] / fb[sel]))),
                         frac_cost_above_reference_by_1e_3=float((cost[sel] > fb[sel] * (1 + 1e-3)).mean()),
                         frac_cost_below_reference_by_1e_3=float((cost[sel] < fb[sel] * (1 - 1e-3)).mean()),
                         same_exception=float(np.mean([errs[k].split(":")[0] == eb[k].split(":")[0] for k in np.flatnonzero(run)])),
                         exceptions=_hist([errs[k] for k in np.flatnonzero(run)]),
                         exits=_hist_codes([codes[k] for k in np.flatnonzero(run)]))
    return out


def g6_report(modes=MODES, limit=None):
    d = np.load(os.path.join(GOLDEN, "g6_reference_vs_itself.npz"))
    ref = g6_reference_side(d)
    dev = g6_device_side(d, modes, limit)
    return dict(
        what="finals of plan_once (expert_planner.py:205-237) on the G6 requests of the 2-D reference map (M = 21): the device in "
             "each arithmetic mode against the real reference's finals (control points, max |dx| / max |x|), next to the "
             "reference under another BLAS kernel set against itself",
        reference_vs_itself=ref, device_vs_reference=dev)


def main():
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import reference_jump          # (the pinned oracle: test infrastructure)
    rep = dict(g1=g1_report(jump_fn=reference_jump), g3_rows=None, g3=None, g6=None)
    rows = g3_report()
    rep["g3_rows"] = rows
    rep["g3"] = g3_summary(rows)
    rep["g6"] = g6_report()
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "ref_fixture_parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(rep, open(path, "w"), indent=1)
    print(json.dumps(dict(g1=rep["g1"], g3=rep["g3"], g6=rep["g6"]), indent=1))


if __name__ == "__main__":
    main()
