"""
Everything a one-GPU `bench.py` run reports BESIDE the driver contract: the sections below extend the
full report that bench.py writes to its side file (`"details"` in the compact line).  Nothing here prints to stdout and
nothing here is inside bench.py's timed region.

    kernel_traffic    HBM bytes per launch of a kernel from the newest COMMITTED rocprofv3 --pmc profile whose command matches
                      the run (PMC counters cannot be read from inside the process); null otherwise
    retries_report    warm_start_plan's chain of <= 5 re-seeded attempts under the bench protocol (expert_planner.py:186-203)
    esdf_report       the ESDF-lookup kernel alone (sample_kernel): the 4096 launch and the launch over a whole step's
                      requests, both dispatch orders, byte conventions, footprint of the field it touches
    cfg1_report       BASELINE.json configs[0]: one plan() of the reference's own shape through the reference-shaped API
    parity_report     finals of every mode against the CPU optimisers, the CPU-vs-CPU controls, the reference-fixture replays
    cpu_leg_extras    (in bench.py's HIP-free CPU child) the controls and cfg1's CPU side

Sections take bench.py's `Rank` object (R): a, ctx, dev, bp, g3, batches, streams, sets, launch(), fence(), ...
"""
import contextlib
import ctypes
import glob
import io
import json
import os
import re
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0
# measured on the MI355X for the ESDF kernel's access shape (tools/gpu_gather_calib.py, profiles/r03_gather_calib.json):
# 32-byte lookups at random offsets of a 432 MB buffer reach 54.6 lookups/ns = 1.75 TB/s of useful bytes, every one a
# 128-byte L2 -> fabric request: 7.0 TB/s of lines.  The same run calibrates FETCH_SIZE for this shape: exactly half of
# the bytes the L2 requests (TCC_EA0_RDREQ_128B x 128), as MI355X_MICROARCH.md states for streaming reads.
GATHER_LINE_ROOFLINE_GBPS = 7020.0


# ================================================================== committed counter profiles
_WORKLOAD_FLAGS = {"--config": "cfg2", "--dtype": "f32x", "--layout": "brick", "--batch": "4096", "--grid": "300",
                   "--waypoints": "20", "--streams": "4", "--batches-per-step": None, "--esdf-order": "spatial"}


def command_workload(cmd):
    """the flags of a `bench.py` command line that change WHICH kernels run on WHAT (defaults filled in); steps, warm-up and
    the report switches do not"""
    toks = cmd.split("bench.py", 1)[-1].split()
    wl = dict(_WORKLOAD_FLAGS)
    for i, t in enumerate(toks):
        if t in wl and i + 1 < len(toks):
            wl[t] = toks[i + 1]
    wl["--planar"] = "--planar" in toks
    wl["--no-order"] = "--no-order" in toks
    wl["--lane-groups"] = "--lane-groups" in toks
    return wl


def run_workload(a):
    """the same flags of THIS run, from its own command line (bench.py keeps it in a.argv)"""
    return command_workload("bench.py " + " ".join(getattr(a, "argv", [])))


def pmc_profile(a, kernel_key, symbol=None):
    """counters per launch of `kernel_key` ("name@workgroups") from the newest committed profiles/*_pmc.json whose COMMAND
    is this run's workload (same config / dtype / layout / sizes / streams) and, when the profile records the kernels'
    symbols and the library can name the one it launched, whose SYMBOL is the run's.  ({}, None) otherwise: a figure
    from another workload or another instantiation is not this run's traffic."""
    want = run_workload(a)
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json")), reverse=True):
        try:
            prof = json.load(open(path))
            if command_workload(prof.get("command", "")) != want:
                continue
            ks = prof["kernels"]
            if kernel_key not in ks:
                continue
            syms = prof.get("kernel_symbols") or {}
            if symbol and kernel_key in syms and not _norm_symbol(syms[kernel_key]).startswith(_norm_symbol(symbol)):
                continue
            return ({c: v["mean_per_dispatch"] for c, v in ks[kernel_key].items() if "mean_per_dispatch" in v},
                    os.path.relpath(path, REPO))
        except Exception:
            continue
    return {}, None


def _norm_symbol(s):
    return re.sub(r"\s+", "", s or "").replace("(anonymousnamespace)::", "")


def hbm_traffic(pm):
    """bytes the L2 moved to and from the fabric (HBM / Infinity Cache) per launch.  From the L2's own read requests by
    size when the profile has them (32 n32 + 64 n64 + 128 n128: exact for every launch shape); otherwise 2 x FETCH_SIZE
    -- the gfx950 correction of MI355X_MICROARCH.md, calibrated for large streaming / gather launches
    (profiles/r03_gather_calib.json; small grids read ~0.9 x there, so the rule is recorded in `traffic_rule`)."""
    if "WRITE_SIZE" not in pm:
        return None
    if "TCC_EA0_RDREQ_128B_sum" in pm and "TCC_EA0_RDREQ_sum" in pm:
        n128, n64, n32 = pm["TCC_EA0_RDREQ_128B_sum"], pm.get("TCC_EA0_RDREQ_64B_sum", 0.0), pm.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        return 128.0 * n128 + 64.0 * n64 + 32.0 * n32 + pm["WRITE_SIZE"] * 1024.0
    if "FETCH_SIZE" not in pm:
        return None
    return (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0


def traffic_rule(pm):
    if "WRITE_SIZE" not in pm:
        return None
    return ("32/64/128-byte L2 read requests (TCC_EA0_RDREQ_*) + WRITE_SIZE" if "TCC_EA0_RDREQ_128B_sum" in pm
            else "2 x FETCH_SIZE + WRITE_SIZE (large-launch calibration)")


def l2_hit(pm):
    if "TCC_HIT_sum" not in pm:
        return None
    return pm["TCC_HIT_sum"] / max(pm["TCC_HIT_sum"] + pm.get("TCC_MISS_sum", 0.0), 1.0)


def kernel_traffic(R, kernel_key):
    sym = R.kernel_symbol(kernel_key.split("@")[0]) if hasattr(R, "kernel_symbol") else None
    pm, src = pmc_profile(R.a, kernel_key, sym)
    return {"traffic": hbm_traffic(pm), "source": src, "rule": traffic_rule(pm), "pm": pm}


# ================================================================== retry chain
def retries_report(R, main_run):
    """the reference's ACCEPTED result is warm_start_plan's: up to five plan_once attempts, the failed ones re-seeded with
    N(0, 0.5) jitter (expert_planner.py:186-203); bench.py's `value` counts first attempts.  Here the whole chain is timed
    under the same protocol, as BatchPlanner.plan runs it on the requests of a step: the first launches of the step's
    batches, then ONE compacted re-launch per attempt of every request that failed so far.  Which requests fail, and their
    re-seeded guesses, are found in an untimed pass (a caller learns them from the status arrays between attempts); the
    timed region replays every launch of the chain, first attempts included, and an attempt's launch waits for every launch
    of the attempt before it (stream events), as it would for the statuses."""
    torch, a, ctx, bp, dev = R.torch, R.a, R.ctx, R.bp, R.dev
    B, M, D, n_sets, n_lanes = R.B, R.M, R.D, R.n_sets, R.n_lanes
    bp._sync()
    R.fence()
    failed_of = lambda st_: ((st_ & 0xff) > 3) | ((st_ & 0x100) != 0)
    t_prep = time.time()
    for bt in R.batches:
        R.launch(bt, bp)
    R.fence()
    head_all = np.concatenate([st_[0] for st_ in R.sets]); tail_all = np.concatenate([st_[1] for st_ in R.sets])
    n_req = B * n_sets
    first_status = torch.stack([bt["status"] for bt in R.batches]).cpu().numpy().reshape(-1)
    todo = np.flatnonzero(failed_of(first_status))
    attempts = np.ones(n_req, dtype=np.int64)
    chain = []

    def launch_retry(e_):
        ctx.set_stream(e_["st"].cuda_stream)
        ctx.check(ctx.lib.neo_optimize_sample_counter(ctx.h, ctypes.c_void_p(e_["nsamp"].data_ptr())))
        ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(e_["order"].data_ptr()), e_["B"]))
        with torch.cuda.stream(e_["st"]):
            bp.optimize_dev(R.g3, e_["x"], e_["head"], e_["tail"], e_["costs"], e_["last"], e_["nit"], e_["nfev"],
                            e_["status"], x0=e_["x0"])

    for att in range(1, 5):
        if todo.size == 0:
            break
        wp_n, ts_n = bp.init_guess(head_all[todo], tail_all[todo], M - 1)
        noise = np.stack([np.random.default_rng([20260, int(i), att]).normal(0.0, 0.5, (D, M - 1)) for i in todo])
        nb_ = int(todo.size)
        st_ = R.streams[att % n_lanes]
        with torch.cuda.stream(st_):
            e_ = dict(B=nb_, st=st_, x0=torch.from_numpy(bp.pack_x(wp_n + noise, ts_n)).to(dev),
                      head=torch.from_numpy(np.ascontiguousarray(head_all[todo])).to(dev),
                      tail=torch.from_numpy(np.ascontiguousarray(tail_all[todo])).to(dev),
                      costs=torch.zeros(nb_, 4, dtype=torch.float64, device=dev), last=torch.zeros(nb_, 4, dtype=torch.float64, device=dev),
                      nit=torch.zeros(nb_, dtype=torch.int32, device=dev), nfev=torch.zeros(nb_, dtype=torch.int32, device=dev),
                      status=torch.zeros(nb_, dtype=torch.int32, device=dev), nsamp=torch.zeros(nb_, dtype=torch.int64, device=dev),
                      order=torch.from_numpy(bp.expected_effort_order(head_all[todo], tail_all[todo], ts_n)).to(dev))
            e_["x"] = torch.empty_like(e_["x0"])
        chain.append(e_)
        launch_retry(e_)
        st_.synchronize()
        attempts[todo] += 1
        todo = todo[failed_of(e_["status"].cpu().numpy())]
    solved_total = n_req - int(todo.size)
    prep_s = time.time() - t_prep

    def chain_step():
        for bt in R.batches:
            R.launch(bt, bp)
        done_prev = []
        for st_ in R.streams:                      # attempt 2 needs the statuses of every first launch
            ev_ = torch.cuda.Event(); ev_.record(st_); done_prev.append(ev_)
        for e_ in chain:
            for ev_ in done_prev:
                e_["st"].wait_event(ev_)
            launch_retry(e_)
            ev_ = torch.cuda.Event(); ev_.record(e_["st"]); done_prev = [ev_]
    R.fence()
    chain_step()
    R.fence()
    k_steps = max(2, a.steps // 4)
    t0 = time.perf_counter()
    for _ in range(k_steps):
        chain_step()
    R.fence()
    el_r = time.perf_counter() - t0
    ctx.set_stream(None)
    rep = {"what": "warm_start_plan for every request of a step (expert_planner.py:186-203; BatchPlanner.plan's chain): the first "
                   "launch of every batch, then ONE compacted re-launch per attempt of the requests that failed so far "
                   "(OverflowError statuses or `collision cost too large`), re-seeded straight line + N(0, 0.5), at most 5 "
                   f"attempts; every launch of the chain inside the timed region, {n_lanes} batches in flight, an attempt waits "
                   "for the attempt before it; the failed sets and their re-seeded guesses come from an untimed pass",
           "max_attempts": 5, "steps": k_steps, "ms_per_step": 1e3 * el_r / k_steps,
           "requests_per_s": n_req * k_steps / el_r,
           "accepted_after_retries_traj_per_s": solved_total * k_steps / el_r,
           "accepted_frac_first_attempt": main_run["accepted_frac"], "accepted_frac_after_retries": solved_total / n_req,
           "mean_attempts": float(attempts.sum()) / n_req, "launches_per_step": n_sets + len(chain),
           "retry_launch_sizes": [int(e_["B"]) for e_ in chain],
           "retry_trajectories_per_step": int(sum(e_["B"] for e_ in chain)), "untimed_preparation_s": prep_s}
    # who is never solved: clearance of the start / goal in the scene's distance field
    if todo.size and getattr(R.g3, "dist", None) is not None:
        rep["never_solved"] = never_solved_profile(R, head_all, tail_all, todo, n_req)
    return rep


def never_solved_profile(R, head_all, tail_all, todo, n_req):
    """requests that fail all five attempts: how far their start and goal are from the nearest obstacle (trilinear field at
    the cell centre nearest the point), against the same figure for the solved ones and the planner's safe distance"""
    dist = np.asarray(R.g3.dist)
    org = np.asarray(R.synth.DOMAIN_ORIGIN)

    def clearance(p):
        i = np.clip(np.floor((p - org) / R.res).astype(np.int64), 0, R.a.grid - 1)
        return dist[i[:, 2], i[:, 1], i[:, 0]]
    solved = np.setdiff1d(np.arange(n_req), todo)
    out = {"n": int(todo.size), "safe_dis": float(R.bp.cfg.safe_dis)}
    for name, idx in (("never_solved", todo), ("solved", solved)):
        cs, cg = clearance(head_all[idx, 0]), clearance(tail_all[idx, 0])
        out[name] = {"start_clearance_median_m": float(np.median(cs)), "goal_clearance_median_m": float(np.median(cg)),
                     "start_or_goal_inside_safe_dis_frac": float(((cs < out["safe_dis"]) | (cg < out["safe_dis"])).mean()),
                     "path_length_median_m": float(np.median(np.linalg.norm(tail_all[idx, 0] - head_all[idx, 0], axis=1)))}
    return out


# ================================================================== one batch, with and without an evaluation budget
def single_batch_budget_report(R, main_run):
    """BASELINE cfg2 taken literally is ONE batch of 4096: its launch lasts as long as its longest run (max nfev against the
    mean).  neo_optimize_batch_budget_dev caps the evaluations per launch and finishes the stragglers in compacted
    re-launches: when is how much of the batch DONE?  Batch 0 alone on the chip, wall clock from the first launch, the host
    round trips for the statuses included; finals bit-identical to the unbudgeted launch (checked here, every trajectory)."""
    torch, ctx, bp, dev, _lib = R.torch, R.ctx, R.bp, R.dev, R._lib
    B, M, D = R.B, R.M, R.D
    bt = R.batches[0]
    nf = main_run["nfev_all"][0]
    state = torch.empty(B * int(ctx.lib.neo_optimize_state_bytes(M, D)), dtype=torch.uint8, device=dev)
    x = torch.empty_like(bt["x0"]); costs = torch.zeros_like(bt["costs"]); last = torch.zeros_like(bt["last"])
    nit = torch.zeros_like(bt["nit"]); nfev = torch.zeros_like(bt["nfev"]); st = torch.zeros_like(bt["status"])
    ctx.set_stream(R.tstream.cuda_stream)
    # (batch 0's expected-effort order: left in bt["order"] by the launch that timed the batch alone, Rank.time_mode)
    ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(bt["order"].data_ptr()) if not R.a.no_order else None, B))
    out = {"what": "batch 0 (4096 requests) alone on the chip: one unbudgeted launch against launches of at most `budget` "
                   "evaluations per trajectory with the stragglers re-launched compacted (BatchPlanner.optimize_budgeted_dev); "
                   "wall clock incl. the status round trips; done_after_first_launch = share of the batch finished when the "
                   "first launch returns",
           "mean_nfev": float(nf.mean()), "max_nfev": int(nf.max())}

    def plain():
        bp.optimize_dev(R.g3, x, bt["head"], bt["tail"], costs, last, nit, nfev, st, x0=bt["x0"])
    for _ in range(2):
        plain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plain()
        torch.cuda.synchronize()
    out["unbudgeted_ms"] = 1e3 * (time.perf_counter() - t0) / 5
    ref = (x.clone(), costs.clone(), last.clone(), nit.clone(), nfev.clone(), st.clone())
    rows = []
    for budget in sorted({int(1.5 * nf.mean()), int(2.5 * nf.mean())}):
        first_ms, total_ms, sizes = [], [], None
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pp = lambda t: ctypes.c_void_p(t.data_ptr())
            ctx.check(ctx.lib.neo_optimize_batch_budget_dev(ctx.h, R.g3.scene_id, B, M, D, pp(bt["x0"]), pp(x), pp(bt["head"]), pp(bt["tail"]),
                                                            pp(costs), pp(last), pp(nit), pp(nfev), pp(st), pp(state), budget, None, 0, 0))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sizes = [B]
            while True:
                sub = torch.nonzero(st == _lib.NEO_TRAJ_SUSPENDED).flatten().to(torch.int32)
                if sub.numel() == 0:
                    break
                ctx.check(ctx.lib.neo_optimize_batch_budget_dev(ctx.h, R.g3.scene_id, B, M, D, pp(bt["x0"]), pp(x), pp(bt["head"]), pp(bt["tail"]),
                                                                pp(costs), pp(last), pp(nit), pp(nfev), pp(st), pp(state), budget, pp(sub),
                                                                int(sub.numel()), 1))
                torch.cuda.synchronize()
                sizes.append(int(sub.numel()))
            t2 = time.perf_counter()
            if rep:
                first_ms.append(1e3 * (t1 - t0)); total_ms.append(1e3 * (t2 - t0))
        same = all(torch.equal(a_, b_) for a_, b_ in zip(ref, (x, costs, last, nit, nfev, st)))
        rows.append({"budget": budget, "first_launch_ms": float(np.mean(first_ms)), "all_done_ms": float(np.mean(total_ms)),
                     "launch_sizes": sizes, "done_after_first_launch": 1.0 - (sizes[1] / B if len(sizes) > 1 else 0.0),
                     "traj_per_s_done_after_first_launch": (B - (sizes[1] if len(sizes) > 1 else 0)) / (float(np.mean(first_ms)) * 1e-3),
                     "bit_identical_to_the_unbudgeted_launch": bool(same)})
    out["budgets"] = rows
    # ---- round 6: results as they finish, from ONE plain launch (neo_optimize_progress_counter): the host polls a counter
    # of finished trajectories through a side stream; when 80 % are in, it copies status and x of the whole batch -- the
    # finished ones are final.  No suspension, no re-launch: the long runs go on undisturbed.
    try:
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        h_cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
        h_st = torch.empty(B, dtype=torch.int32).pin_memory()
        h_x = torch.empty(tuple(x.shape), dtype=torch.float64).pin_memory()
        marks = (0.5, 0.8, 0.9, 0.99, 1.0)
        times = {m_: [] for m_ in marks}
        usable, ok_final = [], True
        for rep in range(4):
            counter.zero_(); st.fill_(-1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bp.optimize_dev(R.g3, x, bt["head"], bt["tail"], costs, last, nit, nfev, st, x0=bt["x0"], progress=counter)
            got_at = {}
            snap_done = None
            with torch.cuda.stream(side):
                while True:
                    h_cnt.copy_(counter, non_blocking=True)
                    side.synchronize()
                    k_ = int(h_cnt[0]); now = time.perf_counter()
                    for m_ in marks:
                        if m_ not in got_at and k_ >= m_ * B:
                            got_at[m_] = now
                            if m_ == 0.8:
                                h_st.copy_(st, non_blocking=True); side.synchronize()       # (status first: tests/test_gpu_budget.py)
                                h_x.copy_(x, non_blocking=True); side.synchronize()
                                got_at["copied"] = time.perf_counter()
                                snap_done = (h_st.numpy() != -1).copy(), h_x.numpy().copy()
                    if k_ >= B:
                        break
            torch.cuda.synchronize()
            if rep:
                for m_ in marks:
                    times[m_].append(1e3 * (got_at[m_] - t0))
                usable.append((float(snap_done[0].mean()), 1e3 * (got_at["copied"] - t0)))
            fin = x.cpu().numpy()
            ok_final = ok_final and bool(np.array_equal(snap_done[1][snap_done[0]], fin[snap_done[0]]))
        ctx.check(ctx.lib.neo_optimize_progress_counter(ctx.h, None))
        same = all(torch.equal(a_, b_) for a_, b_ in zip(ref, (x, costs, last, nit, nfev, st)))
        out["progress"] = {"what": "ONE plain launch of batch 0 with a progress counter (neo_optimize_progress_counter): wall clock from "
                                   "the launch call until the polled counter passes each share of the batch; at 80 % the host copies "
                                   "status and x (D2H, whole arrays) -- `usable_*`: share of the batch final on the host and when",
                           "ms_until_share_finished": {str(m_): float(np.mean(v_)) for m_, v_ in times.items()},
                           "usable_share_on_host": float(np.mean([u_[0] for u_ in usable])),
                           "usable_on_host_ms": float(np.mean([u_[1] for u_ in usable])),
                           "finished_results_were_final": ok_final, "bit_identical_to_the_launch_without_counter": bool(same)}
    except Exception as ex:   # (the side file says what failed; the rest of the report stands)
        out["progress"] = {"error": f"{type(ex).__name__}: {ex}"}
    ctx.set_stream(None)
    bp._sync()
    return out


# ================================================================== the ESDF-lookup kernel on its own
def esdf_report(R):
    """add_sampled_cost + add_sampled_grad_CT (expert_planner.py:392-466) for batch 0 at the initial guess, coefficients
    resident in HBM: sample_kernel outside any timed region, launch durations from HIP events on the kernel's stream"""
    torch, a, ctx, bp, dev, npa, synth, _lib = R.torch, R.a, R.ctx, R.bp, R.dev, R.npa, R.synth, R._lib
    B, M, D, n, n_sets, res, esz = R.B, R.M, R.D, R.n, R.n_sets, R.res, R.esz
    head, tail, wp, ts = R.sets[0]
    b0 = R.batches[0]
    pp = lambda t: ctypes.c_void_p(t.data_ptr())
    coeffs = torch.zeros(B, 6 * M, D, dtype=torch.float64, device=dev)
    cost1 = torch.zeros(B, dtype=torch.float64, device=dev)
    c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
    grad1 = torch.zeros(B, n, dtype=torch.float64, device=dev)
    st1 = torch.zeros(B, dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, R.g3.scene_id, B, M, D, pp(b0["x0"]), pp(b0["head"]), pp(b0["tail"]),
                                              pp(cost1), pp(c4), pp(grad1), pp(coeffs), pp(st1)))
    ns_piece = np.floor(ts / bp.cfg.delta_t).astype(np.int64)
    n_samples = int(ns_piece.sum())
    # S * C * e + 2 n 4 + 20 bytes per evaluation (C = 8 corners of e bytes)
    by_8d2 = n_samples * 8.0 * esz + B * (2 * n * 4 + 20)
    # ... or with what this stand-alone kernel really moves besides the field: fp64 coefficients in, their
    # partials out, durations in / partials out, 2 cost terms
    io_b = 4 if bp.sample_dtype == "f32" else 8    # (round 6: fp32 operand buffers on the fp32 sampling path)
    by_ops = n_samples * 8.0 * esz + B * (2 * 6 * M * D * io_b + M * 8 + M * io_b + 16)
    # footprint of the field: distinct 128-byte lines the launch's lookups touch (linear voxel order)
    cf = coeffs.cpu().numpy().reshape(B, M, 6, D)
    jmax = int(ns_piece.max())
    tj = (np.arange(jmax) * bp.cfg.delta_t)[None, None, :]                                   # [1,1,J]
    pw = np.stack([tj ** k for k in range(6)], axis=-1)                                       # [1,1,J,6]
    pos = np.einsum("bmkd,xyjk->bmjd", cf, pw)                                                # [B,M,J,D]
    valid = np.arange(jmax)[None, None, :] < ns_piece[:, :, None]
    u = (pos[valid] - np.asarray(synth.DOMAIN_ORIGIN)) / res - 0.5
    inside = ((u >= -0.5) & (u < a.grid - 0.5)).all(axis=1)
    i0 = np.clip(np.floor(u[inside]).astype(np.int64), 0, a.grid - 2)
    ids = []
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                ids.append((((i0[:, 2] + dz) * a.grid + i0[:, 1] + dy) * a.grid + i0[:, 0] + dx) * esz // 128)
    footprint = int(np.unique(np.concatenate(ids)).size) * 128
    whole = n_sets > 1 and R.init is None
    if whole:
        Ba = B * n_sets
        coeffs_a = torch.zeros(Ba, 6 * M, D, dtype=torch.float64, device=dev)
        for r_, bt in enumerate(R.batches):
            ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, R.g3.scene_id, B, M, D, pp(bt["x0"]), pp(bt["head"]),
                                                      pp(bt["tail"]), pp(cost1), pp(c4), pp(grad1),
                                                      pp(coeffs_a[r_ * B:(r_ + 1) * B]), pp(st1)))
        ts_a = np.ascontiguousarray(np.concatenate([st_[3] for st_ in R.sets], axis=0))
        head_a = np.concatenate([st_[0] for st_ in R.sets]); tail_a = np.concatenate([st_[1] for st_ in R.sets])
        d_ts_a = torch.from_numpy(ts_a).to(dev)
        ns_a = int(np.floor(ts_a / bp.cfg.delta_t).astype(np.int64).sum())
        by_a = ns_a * 8.0 * esz + Ba * (2 * n * 4 + 20)
    d_ts = torch.from_numpy(np.ascontiguousarray(ts)).to(dev)

    # round 6: the fp32 sampling path takes fp32 coefficient / partials buffers (neo_sampled_terms_batch_f32_dev)
    io32 = bp.sample_dtype == "f32"
    io_dt = torch.float32 if io32 else torch.float64
    sample_fn = ctx.lib.neo_sampled_terms_batch_f32_dev if io32 else ctx.lib.neo_sampled_terms_batch_dev
    K_BACK_TO_BACK = 20

    def time_sample(scene, nb, co, dts, order_np, reps):
        """launch durations of sample_kernel over nb trajectories from HIP events on the kernel's stream: the mean over K
        launches back to back between ONE event pair (`us`: what the kernel costs in a stream of work, and what rocprofv3's
        kernel trace agrees with) and one launch per event pair (the context's profile scope; up to round 5 the only figure:
        ~2 us of event overhead at the 4096 launch)"""
        co = co.to(io_dt)
        c2 = torch.zeros(nb, 2, dtype=torch.float64, device=dev)
        gC = torch.zeros(nb, 6 * M, D, dtype=io_dt, device=dev)
        gT = torch.zeros(nb, M, dtype=io_dt, device=dev)
        od = torch.from_numpy(order_np).to(dev) if order_np is not None else None
        ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, pp(od) if od is not None else None, 1, nb))
        run = lambda: ctx.check(sample_fn(ctx.h, scene, nb, M, D, pp(co), pp(dts), pp(c2), pp(gC), pp(gT)))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ctx.check(ctx.lib.neo_profile_reset(ctx.h))
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
        nl, ms = R.kernel_time(_lib.NEO_KERNEL_ESDF_SAMPLE)
        stream = torch.cuda.current_stream()
        groups = max(reps // K_BACK_TO_BACK, 1)
        us_k = 0.0
        for _ in range(groups):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(K_BACK_TO_BACK):
                run()
            e1.record(stream)
            torch.cuda.synchronize()
            us_k += 1e3 * e0.elapsed_time(e1) / K_BACK_TO_BACK
        ctx.check(ctx.lib.neo_sampled_terms_dispatch_order(ctx.h, None, 0, 0))
        return us_k / groups, nl + groups * K_BACK_TO_BACK, (c2, gC, gT), 1e3 * ms / max(nl, 1)

    def block(scene, layout_name, default_wl):
        """the ESDF-lookup kernel on one field: the 4096-trajectory launch and the launch over every request batch of a
        step, in the chosen dispatch order, the other order timed beside it (same bits either way: checked)"""
        orders = {"index": None, "spatial": npa.BatchPlanner.spatial_order(head, tail)}
        other = "index" if a.esdf_order == "spatial" else "spatial"
        us, nl, out_main, us_single = time_sample(scene, B, coeffs, d_ts, orders[a.esdf_order], 60)
        us_o, _, out_o, _ = time_sample(scene, B, coeffs, d_ts, orders[other], 20)
        same = all(torch.equal(x_, y_) for x_, y_ in zip(out_main, out_o))
        tr = kernel_traffic(R, f"sample_kernel@{B}") if default_wl else {"traffic": None, "source": None, "rule": None, "pm": {}}
        pm_s = tr["pm"]
        e = {"kernel": "sample_kernel", "layout": layout_name, "dispatch_order": a.esdf_order, "bound": "hbm", "trajectories": B,
             "achieved": by_8d2 / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS,
             "unit": "GB/s", "frac": by_8d2 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
             "frac_8d2": by_8d2 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
             "frac_with_operands": by_ops / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
             "kernel_us": us, "launches": nl, "timing": f"HIP events on the kernel's stream, mean over groups of {K_BACK_TO_BACK} launches back to back",
             "kernel_us_one_launch_per_event_pair": us_single,
             "frac_one_launch_per_event_pair": by_8d2 / (us_single * 1e-6) / 1e9 / HBM_PEAK_GBPS,
             "operand_buffers": "f32" if io32 else "f64",
             f"kernel_us_{other}_order": us_o, "orders_give_the_same_bits": bool(same),
             "samples_per_launch": n_samples,
             "algorithmic_bytes_per_launch": by_8d2, "bytes_per_launch_with_operands": by_ops,
             "lookups_per_s": n_samples / (us * 1e-6),
             "esdf_footprint_bytes": footprint, "esdf_bytes": a.grid ** 3 * esz,
             "traffic": tr["traffic"], "l2_hit_rate": l2_hit(pm_s), "traffic_source": tr["source"], "traffic_rule": tr["rule"]}
        if e["traffic"]:
            # the kernel against what it really moves: 128-byte lines for 32-byte lookups
            e["traffic_GBps"] = e["traffic"] / (us * 1e-6) / 1e9
            e["traffic_over_algorithmic"] = e["traffic"] / by_8d2
            e["frac_traffic_of_hbm_peak"] = e["traffic_GBps"] / HBM_PEAK_GBPS
            e["frac_traffic_of_gather_roofline"] = e["traffic_GBps"] / GATHER_LINE_ROOFLINE_GBPS
            e["gather_roofline"] = {"GBps_of_128B_lines": GATHER_LINE_ROOFLINE_GBPS, "source": "profiles/r03_gather_calib.json",
                                    "what": "random 32-byte lookups (two adjacent 16-byte loads per lane) over a 432 MB buffer, "
                                            "the rate the chip sustains for this access shape"}
            lines = pm_s.get("TCC_EA0_RDREQ_128B_sum") or (pm_s.get("FETCH_SIZE", 0) * 1024.0 * 2 / 128)
            e["lookups_per_fetched_line"] = n_samples / max(lines, 1.0)
        # the same kernel over ALL request batches of a step in one launch (n_sets * B trajectories): with more
        # wavefronts than the chip holds at once the launch is bound by throughput, not by the run time of one wavefront
        if whole:
            orders_a = {"index": None, "spatial": npa.BatchPlanner.spatial_order(head_a, tail_a)}
            us_a, nl_a, _, _ = time_sample(scene, Ba, coeffs_a, d_ts_a, orders_a[a.esdf_order], 20)
            us_ao, _, _, _ = time_sample(scene, Ba, coeffs_a, d_ts_a, orders_a[other], 8)
            w_ = {"trajectories": Ba, "kernel_us": us_a, "launches": nl_a, f"kernel_us_{other}_order": us_ao,
                  "dispatch_order": a.esdf_order, "samples_per_launch": ns_a, "algorithmic_bytes_per_launch": by_a,
                  "achieved": by_a / (us_a * 1e-6) / 1e9, "frac_8d2": by_a / (us_a * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            tr_a = kernel_traffic(R, f"sample_kernel@{Ba}") if default_wl else {"traffic": None, "pm": {}}
            if tr_a["traffic"]:
                pm_a = tr_a["pm"]
                lines_a = pm_a.get("TCC_EA0_RDREQ_128B_sum") or (pm_a.get("FETCH_SIZE", 0) * 1024.0 * 2 / 128)
                w_.update(traffic=tr_a["traffic"], l2_hit_rate=l2_hit(pm_a), traffic_source=tr_a["source"], traffic_rule=tr_a["rule"],
                          traffic_GBps=tr_a["traffic"] / (us_a * 1e-6) / 1e9, traffic_over_algorithmic=tr_a["traffic"] / by_a,
                          lookups_per_fetched_line=ns_a / max(lines_a, 1.0),
                          frac_traffic_of_hbm_peak=tr_a["traffic"] / (us_a * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                          frac_traffic_of_gather_roofline=tr_a["traffic"] / (us_a * 1e-6) / 1e9 / GATHER_LINE_ROOFLINE_GBPS)
            e["whole_step_launch"] = w_
        return e

    esdf = block(R.g3.scene_id, a.layout, True)
    if a.layout != "brick" and a.config in ("cfg2", "cfg5") and not a.planar:
        # the same launches on the corner-brick layout of the same field (NEO_LAYOUT_BRICK: a 128-byte line per block of
        # 2 x 2 x 2 cells): fewer lines per path -- counters: profiles/r04_*_pmc_esdf_locality_*.json
        d_occ2 = torch.from_numpy(R.occ).to(dev)
        gb = npa.ESDF3D.from_occupancy(d_occ2, res, synth.DOMAIN_ORIGIN, store=R.store, layout="brick", ctx=ctx)
        del d_occ2
        eb = block(gb.scene_id, "brick", False)
        esdf["brick_layout"] = {k: eb[k] for k in ("layout", "dispatch_order", "kernel_us", "frac_8d2", "frac_with_operands",
                                                   "lookups_per_s", "orders_give_the_same_bits") if k in eb}
        esdf["brick_layout"].update({k: v for k, v in eb.items() if k.startswith("kernel_us_")})
        if "whole_step_launch" in eb:
            esdf["brick_layout"]["whole_step_launch"] = {k: v for k, v in eb["whole_step_launch"].items()
                                                         if k in ("trajectories", "kernel_us", "frac_8d2", "dispatch_order")
                                                         or k.startswith("kernel_us_")}
        ctx.check(ctx.lib.neo_esdf_drop(ctx.h, gb.scene_id))
    bp._sync()
    return esdf


# ================================================================== cfg1
def cfg1_report(R):
    """BASELINE.json configs[0]: ONE plan() of the reference's own shape through the reference-shaped API"""
    npa, synth, ctx = R.npa, R.synth, R.ctx
    occ2 = synth.occupancy_2d(3)
    m2 = npa.ESDF(ctx=ctx)
    m2.occupancy_map_cb(synth.OccupancyGridMsg(occ2))
    h2 = np.array([[0.0, 0.0], [0.0, 0.0]]); t2 = np.array([[5.0, 0.3], [0.8, 0.0]])
    pl1 = npa.MinJerkPlanner(npa.PlannerConfig(), ctx=ctx)
    with contextlib.redirect_stdout(io.StringIO()):
        pl1.plan(m2, h2, t2)
        t1 = time.perf_counter()
        for _ in range(20):
            pl1.plan(m2, h2, t2)
        plan_ms = 1e3 * (time.perf_counter() - t1) / 20
        t1 = time.perf_counter()
        for _ in range(20):
            pl1.batch_plan(m2, h2, t2)
        batch_plan_ms = 1e3 * (time.perf_counter() - t1) / 20
    # ... and that shape in batches: 8192 M = 3 replans of the 2-D map per launch, fp64, default kernel and lane groups
    hb, tb, wb, tsb = synth.replan_requests(5, 8192, 2, D=2, length_range=(4.0, 6.0), jitter=0.3)
    batch_rate = {}
    for name_, lg_ in (("default_kernel", False), ("lane_groups", True)):
        bq = npa.BatchPlanner(ctx=ctx, sample_dtype="f64", lane_groups=lg_)
        xq = bq.pack_x(wb, tsb)
        bq.optimize(m2, xq, hb, tb)
        t1 = time.perf_counter()
        for _ in range(3):
            bq.optimize(m2, xq, hb, tb)
        batch_rate[name_] = 3 * 8192 / (time.perf_counter() - t1)
    R.bp._sync()
    return {"what": "one plan() / batch_plan() call of the reference's shape: M = 3, D = 2, 300 x 300 nearest-cell map, fp64 "
                    "(expert_planner.py:62-80, :142-168); scenario of tests/golden g3 / __graft_entry__.smoke()",
            "plan_ms_gpu": plan_ms, "batch_plan_ms_gpu": batch_plan_ms, "plan_nfev": int(pl1.last_nfev),
            "batched_replans_per_s_fp64_host_buffers": batch_rate,
            "batched_note": "8192 replans of this shape per neo_optimize_batch call (host pointers in and out, PCIe "
                            "included): one trajectory per wavefront / eight per wavefront (NEO_FLAG_LANE_GROUPS)",
            "plan_final_cost_gpu": float(pl1.final_cost)}


# ================================================================== parity tables
def parity_report(R, out, mode_runs, cpu_out, arr):
    """final control points / cost of the GPU runs against the CPU optimisers on the same trajectories, next to the CONTROL:
    the CPU optimiser against itself with fp32-rounded sampling and with the coefficients perturbed by one ulp"""
    a, M, D, w = R.a, R.M, R.D, R.w
    nq = D * (M - 1)

    def delta(bt_x, bt_last, bt_nfev, idx, ref_cost, ref_nfev, ref_wp):
        """GPU results of batch 0 against a CPU optimiser's on the trajectories `idx` (cost of the last
        evaluated point, as the reference reports it, expert_planner.py:233; control points = max |dx| / max |x|)"""
        good = np.isfinite(ref_cost)
        lc = (bt_last * w).sum(dim=1).cpu().numpy()[idx][good]
        rel = np.abs(lc - ref_cost[good]) / np.maximum(np.abs(ref_cost[good]), 1e-12)
        same = bt_nfev.cpu().numpy()[idx][good] == ref_nfev[good]
        gw = bt_x[:, :nq].cpu().numpy()[idx][good]
        dx = np.abs(gw - ref_wp[good]).max(axis=1) / np.maximum(np.abs(ref_wp[good]).max(axis=1), 1e-12)
        return {"n": int(good.sum()), "frac_same_nfev": float(same.mean()),
                "control_points_frac_within_1e_4": float((dx <= 1e-4).mean()),
                "control_points_rel_median": float(np.median(dx)),
                "control_points_rel_max_on_runs_with_same_nfev": float(dx[same].max()) if same.any() else None,
                "final_cost_frac_within_1e_4": float((rel <= 1e-4).mean()), "final_cost_rel_median": float(np.median(rel)),
                "final_cost_frac_within_1e_2": float((rel <= 1e-2).mean()),
                "gpu_median_cost": float(np.median(lc)), "cpu_median_cost": float(np.median(ref_cost[good]))}
    par = {"tolerance": "north_star: final control points within 1e-4 relative of the CPU optimiser's"}
    for m_, r_ in mode_runs.items():
        rb = r_["b0"]
        pm_ = {"vs_cpu_native": delta(rb["x"], rb["last"], rb["nfev"], arr["nat_idx"], arr["nat_cost"], arr["nat_nfev"],
                                     arr["nat_wp"]),
               "vs_numpy_port": delta(rb["x"], rb["last"], rb["nfev"], arr["np_idx"], arr["np_cost"], arr["np_nfev"],
                                      arr["np_wp"])}
        out["modes"][m_]["parity"] = pm_
        if m_ == a.dtype:
            par["gpu_timed_mode_vs_cpu_native"] = pm_["vs_cpu_native"]
            par["gpu_timed_mode_vs_numpy_port"] = pm_["vs_numpy_port"]
    par["per_evaluation_and_decision_replay"] = (
        "tests/test_gpu_replay.py: every point every run of a 256-trajectory cfg2 batch evaluates is re-evaluated by "
        "the fp64 CPU oracle (value 4e-5 / gradient 2e-4 in the all-fp32 mode, 2e-5 / 2e-4 mixed, 1e-10 / 1e-8 fp64) "
        "and every L-BFGS-B decision is re-derived on the host from the recorded values; profiles/r0*_replay_*.json")
    # the three modes against the REFERENCE-GENERATED fixtures on the reference's own 2-D map (tools/ref_fixture_parity.py;
    # thresholds: tests/test_gpu_reference_fixtures.py): G6 = 256 plan_once runs of M = 21, share of finals within 1e-4
    # of the real reference's, beside the reference under another BLAS kernel set against itself
    try:
        import ref_fixture_parity as rfp
        g6 = rfp.g6_report()
        g1 = rfp.g1_report()
        g3_runs = rfp.g3_summary(rfp.g3_report())
        keep = ("n", "finals_within_1e_4", "finals_within_1e_2", "cost_within_1e_4", "cost_within_1e_2", "same_nfev",
                "x_rel_median", "cost_rel_median", "mean_nfev", "same_exception", "exceptions", "exits",
                "median_final_cost", "reference_median_final_cost", "cost_ratio_quantiles", "cost_ratio_log_mean",
                "frac_cost_above_reference_by_1e_3", "frac_cost_below_reference_by_1e_3")
        par["vs_reference_fixtures"] = {
            "what": g6["what"],
            "g6_finals_within_1e_4_of_the_reference": {m_: v_["finals_within_1e_4"] for m_, v_ in g6["device_vs_reference"].items()},
            "g6_reference_vs_itself_other_blas_kernels": g6["reference_vs_itself"]["self_agreement_min"],
            "g6": {"reference_vs_itself": g6["reference_vs_itself"],
                   "device_vs_reference": {m_: {k_: v_[k_] for k_ in keep if k_ in v_} for m_, v_ in g6["device_vs_reference"].items()}},
            "g1_per_evaluation": {m_: {k_: v_[k_] for k_ in ("n", "tolerance", "within_tolerance", "cost_max", "cost_median",
                                                              "grad_max", "grad_median", "coeffs_max")} for m_, v_ in g1.items()},
            "g1_note": "evaluations beyond a mode's tolerance sit at discontinuities of the reference objective (nearest-cell "
                       "faces, int(T / delta_t)): tests/test_gpu_reference_fixtures.py holds each to the reference's own jump there",
            "g3_recorded_runs": g3_runs}
        R.bp._sync()
    except Exception as ex:       # (fixtures missing in a stripped checkout: say so, do not fail the bench line)
        par["vs_reference_fixtures"] = {"error": f"{type(ex).__name__}: {ex}"}
    # exit statuses side by side: GPU modes (modes.*.status_hist), cpu_native, each control
    par["exit_status_hist"] = {"order": "CONVERGED_GRAD, CONVERGED_F, ABNORMAL, MAXITER, NUMERIC_RANGE, NONFINITE, BAD_SCENE",
                               "gpu": {m_: np.bincount(r_["status_all"][0] & 0xff, minlength=7).tolist() for m_, r_ in mode_runs.items()},
                               "cpu_native": cpu_out["cpu_native"].get("status_hist"),
                               "controls": {k_: v_.get("status_hist") for k_, v_ in cpu_out.get("parity_control", {}).items()},
                               "note": "GPU rows: batch 0 of the timed run (4096 runs); cpu_native and controls: the runs of "
                                       "batch 0 the CPU finished inside its time budget"}
    par["control"] = cpu_out.get("parity_control")
    par["reading"] = ("the objective is discontinuous (int(T/dt) sample counts): two faithful CPU implementations "
                      "part at the rates under `control`; the GPU rows are to be read against those, not against 1.0")
    return par


def extend(R, out, mode_runs, cpu_out, arr):
    """run every section that applies to this run and attach it to the full report"""
    a = R.a
    one = R.world == 1 and not R.use_dist
    want = lambda name: a.report_sections == "all" or name in a.report_sections.split(",")
    if want("retries") and one and a.config == "cfg2" and R.init is None and R.n_scenes == 1 and not a.no_retries:
        out["accepted_after_retries"] = retries_report(R, mode_runs[a.dtype])
    if want("budget") and one and a.config == "cfg2" and R.init is None and R.n_scenes == 1 and a.layout in ("brick", "linear") and R.store == "f32":
        out["single_batch_budget"] = single_batch_budget_report(R, mode_runs[a.dtype])
    if want("esdf") and R.n_scenes == 1:
        out["esdf_kernel"] = esdf_report(R)
    if want("cfg1") and one and a.config == "cfg2":
        out["cfg1"] = cfg1_report(R)
        if cpu_out and "cfg1" in cpu_out:
            out["cfg1"].update(cpu_out["cfg1"])
    if want("parity") and cpu_out is not None:
        out["parity"] = parity_report(R, out, mode_runs, cpu_out, arr)


# ================================================================== CPU child: controls and cfg1's CPU side
def cpu_leg_extras(out, arrays, cn, onp, nm, x0, head, tail, M, D, sel, cores, cfg):
    cfgp = onp.PlannerParams()
    w = np.asarray(cfgp.weights)
    nq = D * (M - 1)
    # agreement of the two CPU implementations with each other (different solvers of the same system)
    common = np.intersect1d(arrays["np_idx"], sel)
    if len(common):
        ia_ = {int(i): k for k, i in enumerate(arrays["np_idx"])}
        ib_ = {int(i): k for k, i in enumerate(sel)}
        ia = np.array([ia_[int(i)] for i in common]); ib = np.array([ib_[int(i)] for i in common])
        dx = np.abs(arrays["np_wp"][ia] - arrays["nat_wp"][ib]).max(axis=1) / np.abs(arrays["np_wp"][ia]).max(axis=1)
        out["cpu_native"]["vs_numpy_port"] = dict(
            n=int(len(common)), frac_same_nfev=float((arrays["np_nfev"][ia] == arrays["nat_nfev"][ib]).mean()),
            control_points_frac_within_1e_4=float((dx <= 1e-4).mean()), control_points_rel_median=float(np.median(dx)))

    def control(name, **kw):
        o = cn.optimize_batch(nm, x0[sel], head[sel], tail[sel], M, D, params=cn.make_params(**kw), threads=cores)
        c0 = arrays["nat_cost"]
        c1 = (o["costs_last"] * w).sum(axis=1)
        relc = np.abs(c1 - c0) / np.maximum(np.abs(c0), 1e-12)
        dx = np.abs(o["x"][:, :nq] - arrays["nat_wp"]).max(axis=1) / np.maximum(np.abs(arrays["nat_wp"]).max(axis=1), 1e-12)
        return dict(n=int(len(sel)), what=name, status_hist=np.bincount(o["status"] & 0xff, minlength=7).tolist(),
                    collision_flag_frac=float(((o["status"] & 0x100) != 0).mean()),
                    frac_same_nfev=float((o["nfev"] == arrays["nat_nfev"]).mean()),
                    control_points_frac_within_1e_4=float((dx <= 1e-4).mean()), control_points_rel_median=float(np.median(dx)),
                    final_cost_frac_within_1e_4=float((relc <= 1e-4).mean()), final_cost_rel_median=float(np.median(relc)))
    out["parity_control"] = dict(
        cpu_vs_cpu_fp32_sampling=control("cpu_native against itself with the sampled terms in fp32 arithmetic "
                                         "(what the GPU's timed mode does)", sample_f32=True),
        cpu_vs_cpu_coeffs_1ulp=control("cpu_native against itself with every polynomial coefficient perturbed by a "
                                       "relative 2.2e-16 (what any other solver of the same system does)",
                                       coeff_eps=2.2e-16),
        cpu_vs_cpu_all_fp32_like=control("cpu_native against itself with the sampled terms in fp32 arithmetic, the "
                                         "coefficients perturbed by a relative 1e-7 and the gradient entries by 3e-6 (the "
                                         "per-evaluation deviations of the GPU's all-fp32 mode from the fp64 solve)",
                                         sample_f32=True, coeff_eps=1e-7, grad_eps=3e-6))
    if cfg.get("cfg1"):
        # BASELINE.json configs[0]: one plan() of the reference's own shape on one host core
        from neo_planner_amd import synth
        occ2 = synth.occupancy_2d(3)
        o2 = onp.GridESDF(occ2, synth.RES, 300, 300, (0.0, -15.0))
        h2 = np.array([[0.0, 0.0], [0.0, 0.0]]); t2 = np.array([[5.0, 0.3], [0.8, 0.0]])
        ref = onp.OraclePlanner(onp.PlannerParams())
        with contextlib.redirect_stdout(io.StringIO()):
            ref.plan(o2, h2, t2)
            t0 = time.time()
            for _ in range(5):
                ref.plan(o2, h2, t2)
            port_ms = 1e3 * (time.time() - t0) / 5
            nm2 = cn.NativeMap.from_grid2d(o2)
            npl = cn.NativePlanner()
            iw, its = ref.generate_init_variables(h2, t2)
            t0 = time.time()
            for _ in range(50):
                npl.read_planning_conditions(nm2, h2, t2, iw, its)
                npl.plan_once()
            nat_ms = 1e3 * (time.time() - t0) / 50
        out["cfg1"] = dict(plan_ms_cpu_port=port_ms, plan_ms_cpu_native=nat_ms, plan_final_cost_cpu_port=float(ref.final_cost),
                           cpu_note="one host core; port = oracle/minco_np.py (the reference's per-sample Python loops + "
                                    "SciPy L-BFGS-B), native = oracle/cpu_native cost/gradient in C++ under SciPy L-BFGS-B")
