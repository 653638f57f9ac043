"""
Replay parity of whole device runs.

L-BFGS-B with a More'-Thuente search on this objective amplifies round-off tenfold every ~20 evaluations (DESIGN.md
section 3), so a device run and a CPU run from the same x0 part after ~100 evaluations whatever the kernel does.  The
strongest statement such an objective allows is made here instead, for EVERY evaluation of every run of a cfg2 batch
(300^3 fp32 field in the yz-quad layout, requests filling the volume, M = 21, n = 81), in the timed all-fp32 mode
(`f32x`), the mixed mode (`f32`) and the parity mode (`f64`):

 (1) the device records every point it evaluates with the value and gradient it computed there
     (neo_optimize_trace + neo_optimize_trace_xg).  Each of those points is evaluated again by the fp64 CPU oracle in the
     reference's formulation (oracle/cpu_native: expert_planner.py:539-585 as written, banded 6M x 6M solve) and value
     and gradient must agree to the per-evaluation tolerance of the mode -- at every evaluation of the run, not only at
     the initial guess;
 (2) the recorded (f_k, g_k) are fed to the product's L-BFGS-B control flow compiled for the HOST in fp64
     (tests/host_harness/lbfgs_host.cpp:lbfgs_host_replay -- csrc/neo_lbfgs_sm.hpp, pinned to SciPy by
     tests/test_lbfgs_host.py) which must propose the same trial points (to the rounding of the mode's vectors), take
     the same accept / reject / restart decisions, and stop at the same evaluation with the same status.  Host and
     device may part only inside a DEGENERATE line search -- one that has contracted its step by five orders of
     magnitude onto a single point or onto a jump of the objective, where the last bits of f decide; those runs are
     counted (about 1 %), listed in the report and bounded.

Together: every device run is a valid run of expert_planner.py:213-237 on an objective that is within the stated
tolerance of the reference's at every point the run visits.

Exceptions, counted and bounded instead of hidden: the objective is discontinuous where a duration crosses a multiple
of delta_t (int(T / delta_t), :401).  The all-fp32 mode forms T in fp32; an evaluation whose T / delta_t lies within fp32
rounding of an integer may use one sample more or fewer than the fp64 oracle.  Those evaluations are identified by the
sample count the device records, must be rarer than 1e-3 and must each have a duration within 1e-5 of such a boundary.
"""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

SRC = os.path.join(REPO, "tests", "host_harness", "lbfgs_host.cpp")
INC = os.path.join(REPO, "neo-planner_amd", "csrc")

# Per-evaluation tolerances along the WHOLE run (relative).
#   f / f99: value |df| / |f|, every evaluation / 99 % of them.
#   g: gradient max|dg| / G with G = the largest gradient entry the run meets (max_k max|g_k|): the gradient of this
#      objective is a sum of large terms that cancel as the run converges (collision weight 1e4 against smoothness), so
#      the absolute error of an fp32 term stays what it is while max|g_k| itself shrinks by orders of magnitude -- measured
#      against the terms it is made of, not against what is left of their sum.
#   g_own: the same error against the evaluation's own max|g_k|, asserted where that is still >= 1 % of G.
#   x: the host's trial point against the device's, max|dx| / max(1, max|x|); stp: the line-search step; 99 % quantiles.
#   resync: smallest displacement max|x_k - t| / max(1, max|x_k|) from which the host re-derives the direction from the
#      device's trial point (lbfgs_host_replay): the vectors' own precision times 1e6 / 1e4.
TOL = {"f64": dict(f=1e-10, f99=1e-12, g=1e-9, g_own=1e-8, x=1e-9, stp=1e-8, resync=1e-9),
       "f32": dict(f=4e-5, f99=2e-5, g=4e-5, g_own=4e-4, x=1e-9, stp=1e-8, resync=1e-9),
       "f32x": dict(f=2e-4, f99=4e-5, g=2e-4, g_own=3e-3, x=1e-4, stp=1e-3, resync=3e-4)}
B_REPLAY, M_REPLAY, CAP = 256, 21, 768


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hh") / "lbfgs_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", INC, SRC, "-o", so])
    L = ctypes.CDLL(so)
    c_p, c_i, c_d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.lbfgs_host_replay.restype = c_i
    L.lbfgs_host_replay.argtypes = [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_d, c_d, c_i, c_i, c_i, c_i, c_d, c_p, c_p, c_p, c_p,
                                    c_p, c_p, c_p]
    return L


def replay(L, xr, fr, gr, last_est, cr=None, resync=1e-9):
    """host L-BFGS-B on the recorded values: dict(x_dev [E], stp [E], iter [E], nit, nfev, status, overrun, used)"""
    xr = np.ascontiguousarray(xr, dtype=np.float64)
    gr = np.ascontiguousarray(gr, dtype=np.float64)
    fr = np.ascontiguousarray(fr, dtype=np.float64)
    E, n = xr.shape
    xd = np.zeros(E); stp = np.zeros(E); it = np.zeros(E, dtype=np.int32)
    nit = ctypes.c_int(); nfev = ctypes.c_int(); st = ctypes.c_int(); over = ctypes.c_int()
    crp = None if cr is None else np.ascontiguousarray(cr, dtype=np.float64)
    used = L.lbfgs_host_replay(n, E, xr.ctypes.data, fr.ctypes.data, gr.ctypes.data,
                               crp.ctypes.data if crp is not None else None, int(last_est), 1e-4, 1e-4, 20, 15000, 15000, 10, float(resync),
                               xd.ctypes.data, stp.ctypes.data, it.ctypes.data, ctypes.addressof(nit), ctypes.addressof(nfev),
                               ctypes.addressof(st), ctypes.addressof(over))
    return dict(x_dev=xd, stp=stp, iter=it, nit=nit.value, nfev=nfev.value, status=st.value, overrun=over.value, used=used)


def test_replay_harness_reproduces_a_host_run_exactly(harness):
    """CPU self-check of the replay machinery: a run of the host optimiser on the NumPy oracle (a recorded reference
    scenario), recorded point by point, replays with zero deviation and the same counts"""
    from helpers import golden, load
    from test_lbfgs_host import _oracle_objective, host_minimize
    import test_lbfgs_host as tl
    so_lib = harness
    so_lib.dcsrch_host.restype = ctypes.c_int
    n_runs = 0
    for path in golden("g3_trace_*.npz")[:6]:
        d = load(path)
        for r in range(min(int(d["n_runs"]), 2)):
            x0 = d[f"r{r}_x0"]
            M = (len(x0) + 2) // 3
            _, fgc = _oracle_objective(d, x0[:2 * (M - 1)].reshape(2, M - 1), np.zeros(M))
            rec = []

            def fg(x):
                f, g, c = fgc(x)
                rec.append((x.copy(), f, np.array(g, dtype=np.float64), np.array(c, dtype=np.float64)))
                return f, g, c
            try:
                a = host_minimize(so_lib, x0, fg, entry="lbfgs_host_minimize_sm")
            except Exception:
                continue
            if a["status"] == 4 or not rec:
                continue
            args = (np.stack([q[0] for q in rec]), np.array([q[1] for q in rec]), np.stack([q[2] for q in rec]), 0,
                    np.stack([q[3] for q in rec]))
            out = replay(so_lib, *args, resync=0.0)
            assert out["overrun"] == 0 and out["used"] == len(rec) == a["nfev"]
            assert (out["nit"], out["nfev"], out["status"]) == (a["nit"], a["nfev"], a["status"])
            assert out["x_dev"].max() == 0.0
            # ... and with the direction re-derived from the recorded trial points: the same decisions, trial points to
            # round-off
            out = replay(so_lib, *args, resync=1e-9)
            assert out["overrun"] == 0 and (out["nit"], out["nfev"], out["status"]) == (a["nit"], a["nfev"], a["status"])
            assert out["x_dev"].max() < 1e-7
            n_runs += 1
    assert n_runs >= 4


def _traced_run(mode):
    import torch
    import neo_planner_amd as npa
    from neo_planner_amd import _lib, synth
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    occ = synth.occupancy_3d(0, canopy=80)
    g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), synth.RES, synth.DOMAIN_ORIGIN, layout="yz4", ctx=ctx,
                                   want_dist=True)
    B, M, D = B_REPLAY, M_REPLAY, 3
    n = D * (M - 1) + M
    head, tail, wp, ts = synth.replan_requests(4242, B, M - 1, D=3, **synth.VOLUME)
    bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode, waves_per_simd=2 if mode != "f64" else None)   # bench.py's kernels
    bp._sync()
    x0 = bp.pack_x(wp, ts)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    d_x0, d_x = t(x0), torch.empty(B, n, dtype=torch.float64, device=dev)
    d_h, d_t = t(head), t(tail)
    costs = torch.zeros(B, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
    nit = torch.zeros(B, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); st = torch.zeros_like(nit)
    tr = torch.zeros(B, CAP, 4, dtype=torch.float64, device=dev)
    xg = torch.zeros(B, CAP, 2, n, dtype=torch.float64, device=dev)
    ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), CAP))
    ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, ctypes.c_void_p(xg.data_ptr()), CAP))
    bp.optimize_dev(g3, d_x, d_h, d_t, costs, last, nit, nfev, st, x0=d_x0)
    ctx.synchronize()
    ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
    ctx.check(ctx.lib.neo_optimize_trace_xg(ctx.h, None, 0))
    # the traced launch must be THE run: the same launch without tracing gives the same bits
    d_x2 = torch.empty_like(d_x); nf2 = torch.zeros_like(nfev)
    bp.optimize_dev(g3, d_x2, d_h, d_t, costs, last, nit, nf2, st.clone(), x0=d_x0)
    ctx.synchronize()
    assert torch.equal(d_x, d_x2) and torch.equal(nfev, nf2)
    assert torch.equal(d_x0, t(x0)), "x0 must be left untouched by neo_optimize_batch_from_dev"
    return dict(field=g3.dist, head=head, tail=tail, x0=x0, x=d_x.cpu().numpy(), nfev=nfev.cpu().numpy(),
                nit=nit.cpu().numpy(), status=st.cpu().numpy(), trace=tr.cpu().numpy(), xg=xg.cpu().numpy(), M=M, D=D, n=n,
                cfg=bp.cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32x", "f32", "f64"])
def test_every_evaluation_of_every_run_is_a_reference_evaluation_and_every_decision_is_the_host_optimisers(mode, harness):
    from neo_planner_amd import synth
    from oracle import cpu_native as cn
    r = _traced_run(mode)
    tol = TOL[mode]
    M, D, n = r["M"], r["D"], r["n"]
    nq = D * (M - 1)
    nfev, status = r["nfev"], r["status"] & 0xff
    assert nfev.max() <= CAP, nfev.max()
    nm = cn.NativeMap.from_field3d(r["field"], synth.RES, synth.DOMAIN_ORIGIN)
    cfg = r["cfg"]
    rel_f, rel_g, rel_g_own, ns_diff, near_edge, n_eval = [], [], [], 0, [], 0
    dec_bad, xdev_all, stp_rel_all, runs_checked, worst, n_other_reconstruction = [], [], [], 0, [], 0
    for b in range(B_REPLAY):
        E = int(nfev[b])
        xs, gs = r["xg"][b, :E, 0], r["xg"][b, :E, 1]
        fs, stp_d, ns_d, it_d = (r["trace"][b, :E, k] for k in range(4))
        last_bad = status[b] == 4                      # the run ended on an evaluation the reference leaves by OverflowError
        Ec = E - 1 if last_bad else E
        # ---- (1) every evaluated point on the CPU oracle
        ref = cn.eval_points(nm, xs[:Ec], r["head"][b], r["tail"][b], M, D)
        T = (cfg.T_max - cfg.T_min) / (1.0 + np.exp(-xs[:Ec, nq:])) + cfg.T_min
        ns_ref = np.floor(T / cfg.delta_t).sum(axis=1)
        same_ns = ns_ref == ns_d[:Ec]
        ok = (ref["status"] == 0) & np.isfinite(fs[:Ec])
        sel = ok & same_ns
        rel_f.append(np.abs(fs[:Ec][sel] - ref["f"][sel]) / np.abs(ref["f"][sel]))
        gmax = np.abs(ref["grad"][sel]).max(axis=1)
        G = float(np.abs(ref["grad"][ok]).max()) if ok.any() else 1.0
        dg = np.abs(gs[:Ec][sel] - ref["grad"][sel]).max(axis=1)
        rel_g.append(dg / G)
        if len(dg):
            kk = int(np.argmax(dg))
            kidx = np.flatnonzero(sel)[kk]
            worst.append((float(dg[kk] / G), b, int(kidx), gs[kidx].copy(), ref["grad"][kidx].copy()))
        rel_g_own.append((dg / gmax)[gmax >= 1e-2 * G])
        n_eval += int(ok.sum())
        for k in np.flatnonzero(ok & ~same_ns):
            ns_diff += 1
            q = T[k] / cfg.delta_t
            near_edge.append(float(np.abs(q - np.round(q)).min()))
        # ---- (2) the optimiser's decisions, re-derived on the host from the recorded values
        def same(o):
            return (o["overrun"] == 0 and o["used"] == E and o["nfev"] == E and o["nit"] == int(r["nit"][b])
                    and o["status"] == int(status[b]) and np.array_equal(o["iter"][:E], it_d.astype(np.int32)))
        out = replay(harness, xs, fs, gs, 4 if last_bad else 0, resync=tol["resync"])
        runs_checked += 1
        same_dec = same(out)
        if not same_dec and mode == "f32x":
            # the device's direction is known to the host only through fp32-rounded trial points: the other
            # reconstruction -- the host's own two-loop result on the device's pairs -- is as close to it
            out_b = replay(harness, xs, fs, gs, 4 if last_bad else 0, resync=0.0)
            if same(out_b):
                out, same_dec = out_b, True
                n_other_reconstruction += 1
        if not same_dec:
            k0 = int(np.argmax(out["iter"][:min(E, out["used"])] != it_d[:min(E, out["used"])].astype(np.int32))) \
                if out["used"] else 0
            # where the two part: the first evaluation with another iteration counter, else the evaluation at which one of
            # them stopped.  "flat": the values of the line search in progress there agree to 1e-10 relative -- the search
            # has collapsed onto one point and its decisions are made by the last bits of f
            u = min(E, max(out["used"], 1))
            kd = k0 if (u and out["iter"][k0] != int(it_d[k0])) else u - 1
            ls = fs[(it_d == it_d[min(kd, E - 1)]) & (np.arange(E) <= kd) & (np.arange(E) >= kd - 3)]
            flat = bool(len(ls) >= 2 and (np.nanmax(ls) - np.nanmin(ls)) <= 1e-10 * max(abs(np.nanmax(ls)), 1.0))
            dec_bad.append(dict(b=b, E=E, host=(out["nit"], out["nfev"], out["status"], out["overrun"], out["used"]),
                                dev=(int(r["nit"][b]), E, int(status[b])), parts_at=int(kd), flat_line_search=flat,
                                step_there=float(stp_d[min(kd, E - 1)]),
                                degenerate_line_search=bool(flat or (kd > 0 and stp_d[min(kd, E - 1)] <= 1e-5))))
            continue
        xdev_all.append(out["x_dev"][:E])
        with np.errstate(divide="ignore", invalid="ignore"):
            sr = np.abs(out["stp"][1:E] - stp_d[1:E]) / np.abs(stp_d[1:E])
        stp_rel_all.append(sr[np.isfinite(sr)])
    dump = os.environ.get("NEO_REPLAY_REPORT")
    if dump:
        # raw material for looking at the exceptions offline: the traces of the runs whose decisions differ, and the
        # evaluations with the largest gradient deviation
        os.makedirs(dump, exist_ok=True)
        keep = {}
        for e in dec_bad[:16]:
            b = e["b"]; E = e["E"]
            keep[f"run{b}_x"] = r["xg"][b, :E, 0]; keep[f"run{b}_g"] = r["xg"][b, :E, 1]
            keep[f"run{b}_trace"] = r["trace"][b, :E]; keep[f"run{b}_status"] = status[b]; keep[f"run{b}_nit"] = r["nit"][b]
        for i, (err_, b, k, gg, gc) in enumerate(sorted(worst, key=lambda t: -t[0])[:8]):
            keep[f"worst{i}_b_k_err"] = np.array([b, k, err_]); keep[f"worst{i}_x"] = r["xg"][b, k, 0]
            keep[f"worst{i}_g_dev"] = gg; keep[f"worst{i}_g_cpu"] = gc
            keep[f"worst{i}_head"] = r["head"][b]; keep[f"worst{i}_tail"] = r["tail"][b]
        np.savez_compressed(os.path.join(dump, f"replay_{mode}_cases.npz"), **keep)
    rel_f = np.concatenate(rel_f); rel_g = np.concatenate(rel_g); rel_g_own = np.concatenate(rel_g_own)
    xdev = np.concatenate(xdev_all) if xdev_all else np.zeros(1)
    stp_rel = np.concatenate(stp_rel_all) if stp_rel_all else np.zeros(1)
    q = lambda a: [float(np.quantile(a, p)) for p in (0.5, 0.9, 0.99, 0.999, 1.0)]
    report = dict(mode=mode, runs=runs_checked, evaluations=n_eval, mean_nfev=float(nfev.mean()), max_nfev=int(nfev.max()),
                  status_hist=np.bincount(status, minlength=7).tolist(),
                  value_rel_err_quantiles_50_90_99_999_max=q(rel_f), grad_err_over_run_scale_quantiles=q(rel_g),
                  grad_err_over_own_max_quantiles_where_own_max_ge_1pct_of_run_scale=q(rel_g_own),
                  value_over_tol=int((rel_f > tol["f"]).sum()), grad_over_tol=int((rel_g > tol["g"]).sum()),
                  evaluations_with_other_sample_count=ns_diff, their_distance_to_a_sample_boundary=near_edge,
                  runs_with_identical_decisions=runs_checked - len(dec_bad), runs_with_other_decisions=dec_bad[:40],
                  other_decisions_in_a_flat_line_search=sum(e["flat_line_search"] for e in dec_bad),
                  other_decisions_in_a_degenerate_line_search=sum(e["degenerate_line_search"] for e in dec_bad),
                  runs_reproduced_with_the_hosts_own_direction=n_other_reconstruction,
                  trial_point_dev_quantiles=q(xdev), step_rel_dev_quantiles=q(stp_rel), tolerances=tol)
    dump = os.environ.get("NEO_REPLAY_REPORT")
    if dump:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, f"replay_{mode}.json"), "w") as f:
            json.dump(report, f, indent=1)
    print(json.dumps(report))
    assert n_eval >= 20000
    # (1) per-evaluation parity along the whole run
    assert (rel_f <= tol["f"]).all() and np.quantile(rel_f, 0.99) <= tol["f99"], report["value_rel_err_quantiles_50_90_99_999_max"]
    # gradient: 99.9 % of the evaluations within tolerance; the rest are CELL-FACE events of the fp32 modes -- a sample whose
    # fp32 position falls on the other side of a voxel face than the fp64 oracle's reads the neighbouring cell's gradient
    # (the trilinear interpolant is continuous, its gradient is not), which the collision weight of 1e4 makes visible:
    # rare (<= 2e-4 of the evaluations beyond ten times the tolerance), none in the fp64 mode
    assert np.quantile(rel_g, 0.999) <= tol["g"], report["grad_err_over_run_scale_quantiles"]
    assert (rel_g > 10 * tol["g"]).mean() <= (0.0 if mode == "f64" else 2e-4), int((rel_g > 10 * tol["g"]).sum())
    assert np.quantile(rel_g_own, 0.99) <= tol["g_own"], q(rel_g_own)
    assert ns_diff <= 1e-3 * n_eval and all(e <= 1e-5 for e in near_edge), (ns_diff, near_edge)
    if mode != "f32x":
        assert ns_diff == 0
    # (2) decisions: identical, except where a line search has collapsed onto one point (f flat to 1e-10) and the last bits
    # of f decide -- few, and named.  In the all-fp32 mode the optimiser's own vectors are fp32: the host can re-derive the
    # device's direction from its trial points only to ~1e-4, and a borderline decision may fall the other way.
    # "degenerate": at the evaluation where host and device part, the device's line search has contracted its step below
    # 1e-5 (healthy L-BFGS iterations accept steps near 1) or its values agree to 1e-10: it has collapsed onto a point
    # or sits astride a jump of the objective (a duration crossing a multiple of delta_t), where f changes by 1e-5
    # across 1e-8 in x and the interpolated steps are decided by the last bits.
    healthy = [e for e in dec_bad if not e["degenerate_line_search"]]
    if mode != "f32x":
        assert len(dec_bad) <= 0.05 * runs_checked and not healthy, (len(dec_bad), healthy[:5])
    else:
        assert len(dec_bad) <= 0.15 * runs_checked and len(healthy) <= 0.08 * runs_checked, (len(dec_bad), len(healthy))
    # trial points and steps of the decision-identical runs: 99 % within the rounding of the mode's vectors (the tail is
    # the collapsed line searches again)
    assert np.quantile(xdev, 0.99) <= tol["x"] and np.quantile(stp_rel, 0.99) <= tol["stp"], (q(xdev), q(stp_rel))
