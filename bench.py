#!/usr/bin/env python3
"""
bench.py -- trajectories/sec of the batched replan inner loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a torch.distributed environment launches the N ranks itself (one child process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set for each, rendezvous on 127.0.0.1) before anything touches a GPU.

This file holds the driver contract only: the workload, the timed region, the roofline figure of the dominant kernel
from HIP events, the CPU baseline, and ONE COMPACT JSON line (< 4 KB; `compact_line`, size-tested on the CPU).
Everything else a one-GPU run reports -- the other arithmetic modes' parity tables, the retry chain, the stand-alone
ESDF-lookup kernel's layouts / orders, cfg1, the reference-fixture replays, the CPU-vs-CPU controls -- is produced by
tools/bench_report.py and written to the side file the line names (`"details"`), never to stdout.

Workload (config.workload): BASELINE.json configs[1] -- per GPU one 300^3-voxel fp32 ESDF of a synthetic
random-forest scene (pillars + floating canopy boxes) resident in HBM in the corner-brick layout
(NEO_LAYOUT_BRICK: one 128-byte line per block of 2 x 2 x 2 trilinear cells holding its 27 corners, 432 MB), and request
batches of B = 4096 replans with 20 intermediate waypoints (M = 21 pieces, D = 3, n = 81 variables) whose starts, goals
and waypoints fill the volume (synth.VOLUME).  One launch optimises one batch of 4096 from its initial guess to L-BFGS-B
termination (neo_optimize_batch_dev), inputs already in HBM.  One STEP = one pass of the hot path over
`--batches-per-step` (default 40) different request batches of the scene; launches are issued round-robin on `--streams`
(default 4) HIP streams with separate state and result buffers (`single_batch_*` is one launch alone on the chip).
With N > 1 every rank owns its own scene and batches (weak scaling, no data-path collective); the per-rank results of
every batch are gathered with one RCCL all_gather inside the timed region.

Arithmetic modes: `--dtype f32x` (default, the headline: BASELINE.json's cfg2 is an fp32
configuration) computes everything in fp32; `f32` keeps the coefficient solve, adjoint and optimiser in fp64; `f64` is
the parity mode (`value_parity_mode`).  With one GPU all three are timed in the same run under the same protocol.

The compact line: the driver contract keys, `config`, `roofline` (dominant kernel = optimize_kernel; achieved =
algorithmic bytes per launch -- samples visited x 8 corners x e B + evaluations x (2 n 4 + 20) B, SURVEY.md 8.d2 -- over
the mean launch duration from HIP events on the kernel's stream), `esdf_kernel` (the ESDF-lookup kernel alone),
`cpu_baseline` (the loop-faithful NumPy / SciPy port on the usable host cores, bounded sample), `cpu_native`, `modes`,
`single_batch_traj_per_s`, `details`.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP maps streams onto 4 hardware queues by default; the batches' streams plus RCCL's own then share queues
# and serialise (measured: 359 k instead of 413 k traj/s with the gather on).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
for p_ in (REPO, os.path.join(REPO, "neo-planner_amd"), os.path.join(REPO, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import numpy as np

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak
CANOPY = 80                 # floating boxes per 3-D scene (synth.canopy_boxes)
LINE_LIMIT = 4096           # bytes of the final stdout line (the driver's parser gave up at 20 KB in round 4)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--batches-per-step", type=int, default=None,
                    help="request batches (launches) per step; default 40 for cfg2, 4 for cfg5, 1 otherwise")
    ap.add_argument("--waypoints", type=int, default=20)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--dtype", default="f32x", choices=["f32", "f64", "f32x"],
                    help="f32: fp32 sampled terms, fp64 solve and optimiser; f64: the parity mode; f32x: everything in fp32")
    ap.add_argument("--layout", default="brick", choices=["linear", "yz4", "cell8", "brick"],
                    help="voxel order of the field in HBM (include/neo_planner.h NEO_LAYOUT_*); brick = one 128-byte line per block "
                         "of 2 x 2 x 2 cells (default since round 4), yz4 = the round-2/3 default: one line per 8 cells along x")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall budget of the NumPy-port sample")
    ap.add_argument("--native-seconds", type=float, default=4.0, help="wall budget of each cpu_native run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-retries", action="store_true", help="skip the warm_start_plan retry-chain measurement")
    ap.add_argument("--no-modes", action="store_true", help="time only the --dtype mode (cfg2 on one GPU times all three)")
    ap.add_argument("--no-report", action="store_true",
                    help="the timed region and the compact line only: nothing of tools/bench_report.py runs")
    ap.add_argument("--report-sections", default="all",
                    help="comma list of tools/bench_report.py sections to run: retries, budget, esdf, cfg1, parity (default all); "
                         "profile runs ask for `esdf` alone so that the counters of a launch shape belong to one kind of launch")
    ap.add_argument("--details", default=None,
                    help="where the full report goes (default gpurun_out/bench_details.json, /tmp when that is not writable)")
    ap.add_argument("--esdf-order", default="spatial", choices=["spatial", "index"],
                    help="dispatch order of the stand-alone ESDF-lookup kernel: XCD-aware spatial order "
                         "(BatchPlanner.spatial_order) or index order; the other one is timed beside it")
    ap.add_argument("--planar", action="store_true", help="round-1 workload: every request in the plane z = 2 m, no canopy")
    ap.add_argument("--no-order", action="store_true", help="dispatch trajectories in index order everywhere")
    ap.add_argument("--order", default="auto", choices=["auto", "none", "effort"],
                    help="dispatch order of the PIPELINED steps: index order, or expected effort computed on the device inside the "
                         "timed region (neo_effort_order_dev).  With four launches in flight the order is worth less than its three "
                         "small launches cost (round 6, one box: 1.498 M traj/s without, 1.489 M with); launches that run alone "
                         "(single_batch_*) always use it.  auto: effort when a step has fewer batches than launches in flight "
                         "(cfg3, cfg4: one launch a step runs alone), none otherwise")
    ap.add_argument("--lane-groups", action="store_true",
                    help="small problems (cfg3): eight trajectories per wavefront (NEO_FLAG_LANE_GROUPS)")
    ap.add_argument("--streams", type=int, default=4,
                    help="launches kept in flight per GPU (HIP streams): the tail of a launch -- a few long runs on an "
                         "otherwise idle chip -- overlaps with the next batch")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE.json configs[1..4]; cfg2 is the headline (default).  cfg3: 65536 trajectories, M=3, "
                         "warm-started by the initializer net; cfg4: --scenes scenes x 4096 per GPU; cfg5: 40 waypoints, "
                         "600^3 fp16 field")
    ap.add_argument("--scenes", type=int, default=None,
                    help="cfg4: scenes per GPU; default 256 // max(world, 8) = 32 (BASELINE.json: 256 scenes over 8 GPUs)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to "
                    "exercise the multi-rank code path without GPUs")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: all ranks use cuda:0")
    ap.add_argument("--dry-run", action="store_true",
                    help="testing only (CPU): launcher, rendezvous, gather and timing protocol with the device work "
                         "replaced by a stub; prints value null")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)     # internal: CPU baseline child process
    return ap.parse_args(argv)


# ================================================================== CPU leg (a child process that never touches HIP)
def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


_CPU = {}


def _np_worker(args):
    idx, deadline = args
    from oracle import minco_np as onp
    o3 = onp.Grid3DESDF(_CPU["dist"], _CPU["res"], _CPU["origin"])
    done = []
    for b in idx:
        if time.time() > deadline and done:
            break
        pl = onp.OraclePlanner(onp.PlannerParams())
        pl.read_planning_conditions(o3, _CPU["head"][b], _CPU["tail"][b], _CPU["wp"][b], _CPU["ts"][b])
        try:
            pl.plan_once()
        except Exception:
            pass
        cost = float(np.dot(pl.costs, pl.weights)) if hasattr(pl, "costs") else float("nan")
        nfev = pl.last_result.nfev if pl.last_result is not None else 0
        done.append((int(b), cost, int(nfev), np.asarray(pl.int_wpts, dtype=np.float64).reshape(-1).copy()))
    return done


def cpu_leg(workdir):
    """runs in its own process: (1) single-process calibration and pooled run of the NumPy/SciPy port = `cpu_baseline`,
    (2) cpu_native on the same cores, (3) with `report` set the parity controls and cfg1's CPU side
    (tools/bench_report.py).  Reads workdir/in.npz + field.npy, writes workdir/out.npz + out.json."""
    import multiprocessing as mp
    cfg = json.load(open(os.path.join(workdir, "in.json")))
    d = np.load(os.path.join(workdir, "in.npz"))
    dist = np.load(os.path.join(workdir, "field.npy"), mmap_mode="r")
    head, tail, wp, ts = d["head"], d["tail"], d["wp"], d["ts"]
    B, M = ts.shape
    D = head.shape[2]
    res, origin = cfg["res"], tuple(cfg["origin"])
    cores = usable_cpus()
    out, arrays = {}, {}
    _CPU.update(dist=np.asarray(dist), res=res, origin=origin, head=head, tail=tail, wp=wp, ts=ts)

    # ---- (1) NumPy/SciPy port.  Calibration: two trajectories in this process.
    t0 = time.time()
    cal = _np_worker(([0, 1], time.time() + 1e9))
    single = len(cal) / (time.time() - t0)
    budget = cfg["cpu_seconds"]
    per_worker = max(1, int(budget * single * 1.5) + 1)
    chunks = [(list(range(2 + w, B, cores))[:per_worker], time.time() + budget) for w in range(min(cores, B))]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        res_ = pool.map(_np_worker, chunks)
    dt = time.time() - t0
    done = [r for chunk in res_ for r in chunk]
    pooled = len(done) / dt
    np_all = cal + done
    trusted = pooled / cores >= 0.5 * single
    out["cpu_baseline"] = dict(
        value=pooled if trusted else single, unit="traj/s", cores=cores if trusted else 1, kind="port",
        sample=(f"{len(done)} trajectories of batch 0, {dt:.1f} s wall, one process per usable CPU ({cores}: affinity mask "
                f"capped by the cgroup quota; os.cpu_count() = {os.cpu_count()})" if trusted else
                f"2 trajectories in one process (the pooled run on {cores} processes reached only {pooled / cores:.3f} "
                f"traj/s per process, under half the single-process rate: not a fair figure, not reported)") +
               "; oracle/minco_np.py: per-sample Python loops + scipy L-BFGS-B, fp64, OMP_NUM_THREADS=1",
        per_core=(pooled / cores) if trusted else single, per_core_single=single, pooled_value=pooled,
        pooled_processes=cores, mean_nfev=float(np.mean([r[2] for r in np_all])))
    arrays["np_idx"] = np.array([r[0] for r in np_all])
    arrays["np_cost"] = np.array([r[1] for r in np_all])
    arrays["np_nfev"] = np.array([r[2] for r in np_all])
    arrays["np_wp"] = np.stack([r[3] for r in np_all])

    # ---- (2) cpu_native
    from oracle import cpu_native as cn
    from oracle import minco_np as onp
    nm = cn.NativeMap.from_field3d(np.asarray(dist, dtype=np.float32), res, origin)
    cfgp = onp.PlannerParams()
    tau = -np.log((cfgp.T_max - cfgp.T_min) / (ts - cfgp.T_min) - 1.0)
    x0 = np.concatenate([wp.reshape(B, -1), tau], axis=1)
    sec = cfg["native_seconds"]
    t0 = time.time()
    cn.optimize_batch(nm, x0[:8], head[:8], tail[:8], M, D, threads=1)
    nat_single = 8 / (time.time() - t0)
    t0 = time.time()
    base = cn.optimize_batch(nm, x0, head, tail, M, D, threads=cores, limit_s=sec)
    dt = time.time() - t0
    sel = np.flatnonzero(base["done"])
    out["cpu_native"] = dict(
        value=len(sel) / dt, unit="traj/s", cores=cores, kind="port", per_core=len(sel) / dt / cores,
        per_core_single=nat_single, mean_nfev=float(base["nfev"][sel].mean()),
        sample=f"{len(sel)} trajectories of batch 0, {dt:.1f} s wall on {cores} threads; oracle/cpu_native: C++ fp64, the "
               "reference's banded 6M x 6M system and per-sample loops, L-BFGS-B control flow of csrc/neo_lbfgs.hpp")
    out["cpu_native"]["status_hist"] = np.bincount(base["status"][sel] & 0xff, minlength=7).tolist()
    out["cpu_native"]["collision_flag_frac"] = float(((base["status"][sel] & 0x100) != 0).mean())
    w = np.asarray(cfgp.weights)
    nq = D * (M - 1)
    arrays["nat_idx"] = sel
    arrays["nat_cost"] = (base["costs_last"][sel] * w).sum(axis=1)
    arrays["nat_nfev"] = base["nfev"][sel]
    arrays["nat_wp"] = base["x"][sel, :nq]
    if cfg.get("report"):
        import bench_report
        bench_report.cpu_leg_extras(out, arrays, cn, onp, nm, x0, head, tail, M, D, sel, cores, cfg)
    np.savez(os.path.join(workdir, "out.npz"), **arrays)
    json.dump(out, open(os.path.join(workdir, "out.json"), "w"))


def run_cpu_leg(a, dist_host, res, origin, head, tail, wp, ts):
    wd = tempfile.mkdtemp(prefix="neo_cpu_")
    np.save(os.path.join(wd, "field.npy"), dist_host)
    np.savez(os.path.join(wd, "in.npz"), head=head, tail=tail, wp=wp, ts=ts)
    json.dump(dict(res=res, origin=list(origin), cpu_seconds=a.cpu_seconds, native_seconds=a.native_seconds,
                   report=not a.no_report, cfg1=(a.config == "cfg2" and not a.no_report)),
              open(os.path.join(wd, "in.json"), "w"))
    env = dict(os.environ)
    env["NEO_NO_TORCH_PRELOAD"] = "1"
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--cpu-leg", wd], env=env)
    out = json.load(open(os.path.join(wd, "out.json")))
    arr = dict(np.load(os.path.join(wd, "out.npz")))
    for f in os.listdir(wd):
        os.unlink(os.path.join(wd, f))
    os.rmdir(wd)
    return out, arr


# ================================================================== self-launch of the N ranks
def self_launch(a, argv):
    """one child process per GPU, started before this process touches any GPU; rank 0 prints the JSON line.
    The children are polled: when one of them dies the others are terminated (they would otherwise sit in a
    collective until the RCCL timeout) and its exit code is returned."""
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(a.gpus):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
        rc = 0
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = abs(code)
                    for q in live:
                        q.terminate()
        if rc != EADDRINUSE_RC:
            return rc
    return rc


EADDRINUSE_RC = 98    # a rank exits with this when the rendezvous port was taken between the probe and rank 0's bind


# ================================================================== the compact line
def _r(v, digits=6):
    """numbers to `digits` significant digits (the line is for a parser and a reader, not for bit-exact storage)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _r(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, details_path=None):
    """the ONE stdout line: the driver contract + roofline + cpu_baseline + the few figures VERDICT r4 item 1 lists, from the
    full report `out`; everything else stays in the side file.  Pure function of `out` (tests/test_bench_line.py builds it
    from a canned report and holds it under LINE_LIMIT bytes)."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "batch_per_launch", "batches_per_step", "pieces", "dims", "esdf_voxels", "layout",
                                 "parallelism", "launches_in_flight_per_gpu"))
    modes = out.get("modes") or {}
    if "f64" in modes:
        # the mode that computes in the reference's own arithmetic and meets its 1e-4 as often as the reference meets itself
        line["value_parity_mode"] = modes["f64"].get("value")
    line.update(_pick(out, ("accepted_traj_per_s", "accepted_frac", "host_visible_traj_per_s", "single_batch_ms", "single_batch_traj_per_s",
                            "single_batch_ms_range")))
    if out.get("flags_or"):
        line["flags_or"] = out["flags_or"]
    sb = out.get("single_batch_budget")
    if sb and sb.get("budgets"):
        b0 = sb["budgets"][0]
        line["single_batch_budget"] = dict(_pick(b0, ("budget", "first_launch_ms", "all_done_ms", "done_after_first_launch",
                                                        "bit_identical_to_the_unbudgeted_launch")),
                                           unbudgeted_ms=sb.get("unbudgeted_ms"))
    pg = (sb or {}).get("progress") or {}
    if "ms_until_share_finished" in pg:
        # one plain launch with a progress counter: when 80 % / all of the batch are finished, when the 80 % are on the host
        line["single_batch_progress"] = {"ms_80pct": pg["ms_until_share_finished"].get("0.8"), "ms_all": pg["ms_until_share_finished"].get("1.0"),
                                         "usable_on_host_ms": pg.get("usable_on_host_ms"), "usable_share": pg.get("usable_share_on_host"),
                                         "finals": pg.get("finished_results_were_final")}
    retries = out.get("accepted_after_retries")
    if retries:
        line["accepted_after_retries_traj_per_s"] = retries.get("accepted_after_retries_traj_per_s")
    rf = out.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_aggregate", "frac_one_launch_alone",
                                  "kernel_ms", "launches", "concurrent_launches", "algorithmic_bytes_per_launch"))
    line["roofline"]["traffic"] = rf.get("traffic")
    line["roofline"]["traffic_source"] = rf.get("traffic_source")
    es = out.get("esdf_kernel")
    if es:
        e = _pick(es, ("kernel", "bound", "achieved", "peak", "unit", "frac", "kernel_us", "kernel_us_one_launch_per_event_pair",
                       "frac_one_launch_per_event_pair", "operand_buffers", "trajectories", "traffic",
                       "traffic_over_algorithmic", "traffic_source", "lookups_per_fetched_line", "l2_hit_rate",
                       "orders_give_the_same_bits"))
        ws = es.get("whole_step_launch")
        if ws:
            e["whole_step_launch"] = _pick(ws, ("trajectories", "kernel_us", "frac_8d2", "traffic_over_algorithmic"))
        line["esdf_kernel"] = e
    eb = out.get("esdf_build")
    if eb:
        line["esdf_build"] = _pick(eb, ("voxels", "ms", "GBps"))
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "per_core"))
        c["sample"] = str(cb.get("sample", ""))[:160]
        line["cpu_baseline"] = c
    cn = out.get("cpu_native")
    if cn:
        line["cpu_native"] = _pick(cn, ("value", "unit", "cores", "kind"))
    ml = {}
    for m_, v_ in modes.items():
        e = _pick(v_, ("value", "single_batch_traj_per_s", "accepted_frac", "roofline_frac"))
        par = (v_.get("parity") or {}).get("vs_cpu_native") or {}
        if "control_points_frac_within_1e_4" in par:
            e["finals_within_1e_4"] = par["control_points_frac_within_1e_4"]
        g6 = ((((out.get("parity") or {}).get("vs_reference_fixtures") or {})
               .get("g6_finals_within_1e_4_of_the_reference")) or {})
        if m_ in g6:
            e["g6_finals_within_1e_4_of_reference"] = g6[m_]
        ml[m_] = e
    if ml:
        line["modes"] = ml
    ctl = (((out.get("parity") or {}).get("control") or {}).get("cpu_vs_cpu_coeffs_1ulp") or {})
    if "control_points_frac_within_1e_4" in ctl:
        line["cpu_vs_cpu_1ulp_finals_within_1e_4"] = ctl["control_points_frac_within_1e_4"]
    c1 = out.get("cfg1")
    if c1:
        line["cfg1"] = _pick(c1, ("plan_ms_gpu", "plan_ms_cpu_port", "plan_ms_cpu_native"))
    line.update(_pick(out, ("rccl_ranks", "gather_ok", "dist_backend", "per_rank_traj_per_s", "timed_region_s")))
    line["details"] = details_path
    line = _r(line)
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: drop optional blocks, last first, until the line fits
    for k in ("cfg1", "esdf_build", "cpu_native", "single_batch_budget", "single_batch_progress", "per_rank_traj_per_s",
              "accepted_after_retries_traj_per_s"):
        if len(s) <= LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:
        line["config"]["workload"] = line["config"].get("workload", "")[:120]
        line.get("cpu_baseline", {}).pop("sample", None)
        s = json.dumps(line, separators=(",", ":"))
    return s


def write_details(a, out):
    """the full report -> a side file; returns its path relative to the repository (or absolute under /tmp)"""
    cands = [a.details] if a.details else [os.path.join(REPO, "gpurun_out", "bench_details.json"),
                                            os.path.join(tempfile.gettempdir(), "neo_bench_details.json")]
    for path in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            ap_ = os.path.abspath(path)
            return os.path.relpath(ap_, REPO) if ap_.startswith(REPO + os.sep) else ap_
        except OSError:
            continue
    return None


# ================================================================== one rank
def workload_sets(a, rank, M, D, n_sets, n_scenes):
    """request batches of this rank: list of (head, tail, wp, ts)"""
    from neo_planner_amd import synth
    kw = {} if a.planar else dict(synth.VOLUME)
    sets = []
    for r in range(n_sets):
        if n_scenes == 1:
            lr = (4.0, 6.0) if a.config == "cfg3" else (10.0, 28.0)   # cfg3: 5 m local targets, like the reference's M = 3
            sets.append(synth.replan_requests(rank + 1000 * r, a.batch, M - 1, D=D, length_range=lr, **kw))
        else:
            parts = [synth.replan_requests(rank * n_scenes + s + 1000 * r, 4096, M - 1, D=D, **kw) for s in range(n_scenes)]
            sets.append(tuple(np.concatenate([p[k] for p in parts]) for k in range(4)))
    return sets


class Rank:
    """everything one rank keeps in HBM for the bench, the launch of one request batch, the fence and the timed region.
    tools/bench_report.py's sections take this object (attributes: a, ctx, dev, bp, g3, batches, streams, sets, ...)."""

    def __init__(self, a, rank, local_rank, world, use_dist):
        import torch
        import neo_planner_amd as npa
        from neo_planner_amd import _lib, synth, sharding
        self.a, self.rank, self.world, self.use_dist = a, rank, world, use_dist
        self.torch, self.npa, self._lib, self.synth, self.sharding = torch, npa, _lib, synth, sharding
        self.store = "f16" if a.config == "cfg5" else "f32"
        self.n_scenes = a.scenes if a.config == "cfg4" else 1
        self.n_sets = a.batches_per_step
        self.M, self.D, self.B = a.waypoints + 1, 3, a.batch
        self.n = self.D * (self.M - 1) + self.M
        self.res = 30.0 / a.grid
        self.canopy = 0 if a.planar else CANOPY
        self.esz = 4 if self.store == "f32" else 2
        M, D, B, n = self.M, self.D, self.B, self.n
        self.t_setup = time.time()
        self.occ = synth.occupancy_3d(rank, n=a.grid, res=self.res, canopy=self.canopy)      # scene = rank (weak scaling)
        self.sets = workload_sets(a, rank, M, D, self.n_sets, self.n_scenes)
        if a.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        self.dev = dev = torch.device("cuda", local_rank)
        self.dist = None
        if use_dist:
            import torch.distributed as dist_
            self.dist = dist_
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            try:
                if a.dist_backend == "nccl":
                    dist_.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                else:
                    dist_.init_process_group(a.dist_backend, rank=rank, world_size=world)
            except Exception as ex:
                if "EADDRINUSE" in str(ex) or "address already in use" in str(ex).lower():
                    sys.exit(EADDRINUSE_RC)      # self_launch picks another port
                raise
        # one explicit (non-default) stream for everything: torch copies, our kernels, RCCL.  The default
        # stream has handle 0, which the C ABI reads as "create your own stream".
        self.tstream = tstream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        assert tstream.cuda_stream != 0
        self.ctx = ctx = npa.Context(local_rank, stream=tstream.cuda_stream)
        self.bp = self.planner_for(a.dtype)
        self.bp._sync()
        self.want_cpu = world == 1 and rank == 0 and not a.no_cpu
        self.g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(self.occ).to(dev), self.res, synth.DOMAIN_ORIGIN, store=self.store,
                                            layout=a.layout, ctx=ctx, want_dist=self.want_cpu)
        self.slots = None
        self.scenes = [self.g3]
        if self.n_scenes > 1:
            # cfg4: every scene's field resident in this GPU's HBM, trajectories carry their scene's table slot
            for s_ in range(1, self.n_scenes):
                o_ = synth.occupancy_3d(rank * self.n_scenes + s_, n=a.grid, res=self.res, canopy=self.canopy)
                self.scenes.append(npa.ESDF3D.from_occupancy(torch.from_numpy(o_).to(dev), self.res, synth.DOMAIN_ORIGIN,
                                                             store=self.store, layout=a.layout, ctx=ctx))
            sl = [ctx.lib.neo_scene_slot(ctx.h, sc.scene_id) for sc in self.scenes]
            self.slots = torch.tensor(np.repeat(sl, 4096), dtype=torch.int32, device=dev)
        self.init = None
        self.init_report = None
        if a.config == "cfg3":
            self._setup_initializer()
        self.w = torch.tensor(self.bp.cfg.weights, dtype=torch.float64, device=dev)
        # `--streams` launches in flight; every request batch has its own inputs, state and result buffers and always runs
        # on the same stream (batch r -> stream r mod streams), so reuse of its buffers is ordered by that stream
        self.n_lanes = max(1, a.streams)
        self.streams = [tstream] + [torch.cuda.Stream(device=dev) for _ in range(self.n_lanes - 1)]
        self.batches = []
        for r, (h_, t_, wp_, ts_) in enumerate(self.sets):
            st_ = self.streams[r % self.n_lanes]
            with torch.cuda.stream(st_):
                x0 = torch.from_numpy(self.bp.pack_x(wp_, ts_)).to(dev)
                # (round 6: the dispatch order -- runs expected to be long first -- is computed on the device from the batch's own
                #  buffers at every launch, inside the timed region: `effort_order_dev`; up to round 5 a host argsort at set-up)
                self.batches.append(dict(
                    st=st_, x0=x0, x=torch.empty_like(x0), head=torch.from_numpy(h_).to(dev), tail=torch.from_numpy(t_).to(dev),
                    order=torch.zeros(B, dtype=torch.int32, device=dev),
                    okeys=torch.zeros(int(ctx.lib.neo_effort_order_scratch_bytes(B)) // 8 + 1, dtype=torch.float64, device=dev), nsamp=torch.zeros(B, dtype=torch.int64, device=dev),
                    # pinned host mirrors of what a caller reads back (`host_visible_traj_per_s`: time_mode(d2h=True))
                    h_x=torch.empty(B, n, dtype=torch.float64).pin_memory(), h_status=torch.empty(B, dtype=torch.int32).pin_memory(),
                    costs=torch.zeros(B, 4, dtype=torch.float64, device=dev), last=torch.zeros(B, 4, dtype=torch.float64, device=dev),
                    nit=torch.zeros(B, dtype=torch.int32, device=dev), nfev=torch.zeros(B, dtype=torch.int32, device=dev),
                    status=torch.zeros(B, dtype=torch.int32, device=dev),
                    gathered=torch.empty(world * B, n + 5, dtype=torch.float32, device=dev) if use_dist else None))

    def planner_for(self, mode):
        # several launches in flight -> the throughput variant of the optimiser kernel (two wavefronts per SIMD)
        p_ = self.npa.BatchPlanner(ctx=self.ctx, sample_dtype=mode, waves_per_simd=2 if self.a.streams > 1 else None,
                                   lane_groups=self.a.lane_groups)
        p_.flags |= int(os.environ.get("NEO_BENCH_FLAGS_OR", "0"))     # kernel experiments
        return p_

    def _setup_initializer(self):
        """cfg3: initializer warm start (random weights: the reference's trained ones are not in its tree).  One synthetic
        depth image per scene through the backbone once; the dense head runs per trajectory inside the timed region."""
        torch, synth, a, dev, B = self.torch, self.synth, self.a, self.dev, self.B
        from neo_planner_amd import initializer as ini
        head, tail = self.sets[0][0], self.sets[0][1]
        torch.manual_seed(1234 + self.rank)
        self.init = init = ini.BatchInitializer(device=dev)
        # pinhole depth image of the scene's boxes from the mean start pose, looking along +x
        depth = ini.raycast_depth(synth.forest_boxes(self.rank), synth.canopy_boxes(self.rank, self.canopy) if self.canopy else [],
                                  eye=head[:, 0].mean(axis=0))
        t_bb = time.perf_counter()
        self.feat = init.scene_feature(depth)
        torch.cuda.synchronize()
        first = 1e3 * (time.perf_counter() - t_bb)
        t_bb = time.perf_counter()
        for _ in range(5):
            self.feat = init.scene_feature(depth)
        torch.cuda.synchronize()
        self.init_report = {"backbone_ms_per_scene": 1e3 * (time.perf_counter() - t_bb) / 5, "backbone_ms_first_call": first,
                            "weights": "random (the reference's trained weights are not in its tree): parity unpinned"}
        goal_dir = tail[:, 0] - head[:, 0]
        motion = np.concatenate([head[:, 1], np.tile(np.eye(3).reshape(-1), (B, 1)), np.zeros((B, 3)), head[:, 1],
                                 goal_dir, tail[:, 1]], axis=1)
        self.d_motion = torch.from_numpy(motion.astype(np.float32)).to(dev)
        self.d_R = torch.eye(3, dtype=torch.float64, device=dev).expand(B, 3, 3).contiguous()
        self.d_p0 = torch.from_numpy(head[:, 0]).to(dev)
        self.line = torch.from_numpy(np.stack([head[:, 0] + goal_dir * f for f in (1 / 3, 2 / 3)], axis=2)).to(dev)   # [B,3,2]

    def warm_start(self, x0_):
        """network output -> x0: body-frame waypoints (a small correction on the straight line, the net being untrained)
        and durations clamped into (T_min, T_max), then tau = map_T2tau(ts)"""
        torch, B, M, D = self.torch, self.B, self.M, self.D
        T_lo, T_hi = self.bp.cfg.T_min, self.bp.cfg.T_max
        out = self.init.net.head(self.feat, self.d_motion).double()
        local = out[:, :6].reshape(B, 2, 3)
        world_ = torch.einsum("bij,bwj->bwi", self.d_R, local) + self.d_p0[:, None, :]
        wp_ = self.line + 0.05 * (world_.transpose(1, 2) - self.d_p0[:, :, None])
        eps = 1e-3 * (T_hi - T_lo)
        ts_ = (2.5 + out[:, 6:]).clamp(T_lo + eps, T_hi - eps)
        tau_ = -torch.log((T_hi - T_lo) / (ts_ - T_lo) - 1.0)
        x0_[:, :D * (M - 1)] = wp_.reshape(B, -1)
        x0_[:, D * (M - 1):] = tau_

    def effort_order_dev(self, bt):
        """BatchPlanner.expected_effort_order on the device, from the batch's resident x0 / head / tail: time slack of the
        guess, sum(ts) v_max / distance, largest first (list scheduling: a long run that starts last is the launch's tail).
        neo_effort_order_dev: a keys kernel and a radix sort on the batch's stream (a first version with torch -- sigmoid, norm, a
        stable argsort: ~20 launches a batch -- cost the step 7 %)"""
        ctx, pp = self.ctx, (lambda t: ctypes.c_void_p(t.data_ptr()))
        ctx.check(ctx.lib.neo_effort_order_dev(ctx.h, self.B, self.M, self.D, pp(bt["x0"]), pp(bt["head"]), pp(bt["tail"]),
                                               pp(bt["okeys"]), pp(bt["order"])))

    def launch(self, bt, bpm, d2h=False, ordered=None):
        """ordered: dispatch the batch's workgroups in expected-effort order, computed on the device ahead of the launch
        (None: `--order`'s choice for the pipelined steps)"""
        torch, ctx, B = self.torch, self.ctx, self.B
        if ordered is None:
            ordered = (self.a.order == "effort" or (self.a.order == "auto" and self.n_sets < self.n_lanes)) and not self.a.no_order
        else:
            ordered = ordered and not self.a.no_order
        ctx.set_stream(bt["st"].cuda_stream)
        ctx.check(ctx.lib.neo_optimize_sample_counter(ctx.h, ctypes.c_void_p(bt["nsamp"].data_ptr())))
        with torch.cuda.stream(bt["st"]):
            if self.init is not None:
                with torch.no_grad():
                    self.warm_start(bt["x0"])
        if ordered:
            self.effort_order_dev(bt)        # (on the batch's stream, ahead of the launch that reads bt["order"])
        ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h, ctypes.c_void_p(bt["order"].data_ptr()) if ordered else None, B))
        with torch.cuda.stream(bt["st"]):
            # start points are read from x0, results written to x: no copy per launch (neo_optimize_batch_from_dev)
            bpm.optimize_dev(self.g3, bt["x"], bt["head"], bt["tail"], bt["costs"], bt["last"], bt["nit"], bt["nfev"],
                             bt["status"], slots=self.slots, x0=bt["x0"])
            if d2h:
                bt["h_x"].copy_(bt["x"], non_blocking=True)
                bt["h_status"].copy_(bt["status"], non_blocking=True)
            if self.use_dist:
                # results to every rank: final x, total cost, 4 cost terms.  The gather runs
                # behind the batch on the process group's own stream; this batch's stream does not wait for it
                # (the fence at the end of the timed region does), only the batch's next use of its buffers does.
                if bt.get("work") is not None:
                    bt["work"].wait()
                bt["packed"] = self.sharding.pack_results(bt["x"], bt["costs"], self.bp.cfg.weights, ctx=self.ctx,
                                                          out=bt.get("packed"))
                _, bt["work"] = self.sharding.gather_results(bt["packed"], self.world, out=bt["gathered"], force=True,
                                                             async_op=True)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def kernel_symbol(self, name):
        """the instantiation this run launches for kernel family `name`, as rocprofv3 spells it (demangled, up to the argument
        list): what a committed counter profile's `kernel_symbols` entry must say for its figures to be this run's
        (tools/bench_report.py pmc_profile).  None for the families the bench has no traffic figure for."""
        a, M, D = self.a, self.M, self.D
        lay = {"linear": 0, "yz4": 1, "cell8": 2, "brick": 3}[a.layout]
        elem = "float" if self.store == "f32" else "__half"
        real = "double" if a.dtype == "f64" else "float"
        num = "float" if a.dtype == "f32x" else "double"
        lookup = f"neo::Lookup3D<{real}, {elem}, {lay}>"
        if name == "sample_kernel":
            io = "double" if a.dtype == "f64" else "float"      # (round 6: fp32 operand buffers on the fp32 sampling path)
            return f"void neo::sample_kernel<{D}, {real}, neo::Map3D, {lookup}, {io}>"
        if name == "optimize_kernel":
            ns = min(max((self.n + 63) // 64, 1), 4)
            lg = f"neo::WaveLanesPD<{D}>" if D * M <= 64 else "neo::WaveLanes"
            return f"void neo::optimize_kernel<{D}, {ns}, {real}, neo::Map3D, {lookup}, 2, {lg}, {num}, false>"
        return None

    def kernel_time(self, which):
        """(launches, total ms) of a kernel family since neo_profile_reset: HIP events on the stream each launch ran on"""
        launches, kms = ctypes.c_int64(), ctypes.c_double()
        self.ctx.check(self.ctx.lib.neo_profile_read(self.ctx.h, which, ctypes.byref(launches), ctypes.byref(kms)))
        return int(launches.value), float(kms.value)

    def time_mode(self, mode, d2h=False):
        """W warm-up steps, then exactly K timed steps of the hot path in arithmetic mode `mode`, fenced by a barrier
        and a device synchronisation on both sides; returns the timing and what the launches of the last step left.
        d2h: every launch is followed by the copy of its final points and statuses to pinned host memory (what a host-side
        caller sees; returns only the elapsed time)"""
        torch, ctx, a = self.torch, self.ctx, self.a
        bpm = self.bp if mode == a.dtype else self.planner_for(mode)
        bpm._sync()
        self.fence()
        for k in range(a.warmup):
            for bt in self.batches:
                self.launch(bt, bpm, d2h)
        self.fence()
        ctx.check(ctx.lib.neo_profile_reset(ctx.h))
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
        t0 = time.perf_counter()
        for k in range(a.steps):
            for bt in self.batches:
                self.launch(bt, bpm, d2h)
        self.fence()
        el = time.perf_counter() - t0
        ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
        ctx.set_stream(None)
        if d2h:
            return dict(mode=mode, elapsed=el)
        launches, kms = self.kernel_time(self._lib.NEO_KERNEL_OPTIMIZE)
        # one launch ALONE on the chip (outside the timed region): what a caller with a single request batch gets -- with
        # `--streams` launches in flight each one lasts longer than it would by itself
        # A launch alone lasts as long as its LONGEST run (one wavefront, ~600 evaluations of a mean of 135), and which run
        # that is changes with the last bit of the arithmetic: the figure is the mean over the step's first eight request
        # batches, each launched by itself (round 5; batch 0 alone moved 6.05 -> 6.99 ms with a change that made every
        # evaluation faster), the range beside it.
        solo, solo_range = None, None
        if self.rank == 0 and not self.use_dist:
            each = []
            for bt in self.batches[:8]:
                ctx.check(ctx.lib.neo_profile_reset(ctx.h))
                ctx.check(ctx.lib.neo_profile_enable(ctx.h, 1))
                self.launch(bt, bpm, ordered=True)      # (a launch alone: long runs first, or the longest one is its tail)
                self.fence()
                ctx.check(ctx.lib.neo_profile_enable(ctx.h, 0))
                l_s, m_s = self.kernel_time(self._lib.NEO_KERNEL_OPTIMIZE)
                each.append(m_s / max(l_s, 1))
            ctx.set_stream(None)
            solo, solo_range = sum(each) / len(each), [min(each), max(each)]
        nfev_all = torch.stack([bt["nfev"] for bt in self.batches]).cpu().numpy().astype(np.int64)
        nsamp_all = torch.stack([bt["nsamp"] for bt in self.batches]).cpu().numpy()
        status_all = torch.stack([bt["status"] for bt in self.batches]).cpu().numpy()
        # results the reference accepts: L-BFGS-B ended on its own (converged / abnormal line search: SciPy returns
        # either without raising) and the weighted collision cost is within tolerance (expert_planner.py:235-237)
        accepted = ((status_all & 0xff) <= 2) & ((status_all & 0x100) == 0)
        # algorithmic bytes of ONE launch, mean over the step's batches: S*C*e per evaluation + 2*n*4 + 20
        bytes_launch = (float(nsamp_all.sum()) * 8 * self.esz + float(nfev_all.sum()) * (2 * self.n * 4 + 20)) / self.n_sets
        b0_ = self.batches[0]
        return dict(mode=mode, elapsed=el, kernel_ms=kms / max(launches, 1), launches=launches, nfev_all=nfev_all, solo_ms=solo, solo_ms_range=solo_range,
                    nsamp_all=nsamp_all, status_all=status_all, accepted_frac=float(accepted.mean()),
                    bytes_launch=bytes_launch, mean_nit=float(b0_["nit"].float().mean().item()),
                    b0=dict(x=b0_["x"].clone(), last=b0_["last"].clone(), nfev=b0_["nfev"].clone()))

    def mode_line(self, r_):
        """one arithmetic mode under the bench protocol (this rank; with N ranks `value` is the job's)"""
        a, B = self.a, self.B
        v_ = B * self.n_sets * a.steps / r_["elapsed"]
        ach = r_["bytes_launch"] / (r_["kernel_ms"] * 1e-3) / 1e9
        return {"value": v_, "unit": "traj/s", "ms_per_step": 1e3 * r_["elapsed"] / a.steps,
                "accepted_frac": r_["accepted_frac"], "accepted_traj_per_s": v_ * r_["accepted_frac"],
                "single_batch_ms": r_["solo_ms"],
                "single_batch_traj_per_s": (B / (r_["solo_ms"] * 1e-3)) if r_["solo_ms"] else None,
                "kernel_ms": r_["kernel_ms"], "roofline_frac": ach / HBM_PEAK_GBPS,
                "roofline_frac_aggregate": r_["bytes_launch"] * self.n_sets * a.steps / r_["elapsed"] / 1e9 / HBM_PEAK_GBPS,
                "mean_nfev": float(r_["nfev_all"].mean()), "max_nfev": int(r_["nfev_all"].max()),
                "status_hist": np.bincount(r_["status_all"].reshape(-1) & 0xff, minlength=7).tolist(),
                "sampling_arithmetic": "f64" if r_["mode"] == "f64" else "f32",
                "solve_and_optimiser_arithmetic": "f32" if r_["mode"] == "f32x" else "f64"}

    def esdf_build_time(self):
        """ESDF construction, occupancy resident in HBM: the same call again, timed"""
        torch, a, ctx = self.torch, self.a, self.ctx
        d_occ_ = torch.from_numpy(self.occ).to(self.dev)
        tb_ = []
        for _ in range(3):
            torch.cuda.synchronize(); t1_ = time.perf_counter()
            tmp_ = self.npa.ESDF3D.from_occupancy(d_occ_, self.res, self.synth.DOMAIN_ORIGIN, store=self.store, layout=a.layout, ctx=ctx)
            torch.cuda.synchronize(); tb_.append(time.perf_counter() - t1_)
            ctx.lib.neo_esdf_drop(ctx.h, tmp_.scene_id)
        nv_ = a.grid ** 3
        # EDT: 1 B occupancy in, 2 B row distances out and in, 4 B plane distances out and in, 4 B fp32 distance out = 17 B
        # (+ 4 B the packing pass reads back); packing writes 1 (linear), 4 (yz4 / brick) or 8 (cell8) stored elements per voxel
        elems_ = {"linear": 1, "yz4": 4, "cell8": 8, "brick": 4}[a.layout]
        bpv_ = 17 + 4 + elems_ * self.esz
        return {"what": "neo_esdf_build_3d: exact EDT of the occupancy grid (three separable integer passes) + layout "
                        "packing, device to device, wall time of the whole call (allocations included)",
                "voxels": nv_, "ms": 1e3 * min(tb_), "algorithmic_bytes_per_voxel": bpv_, "GBps": nv_ * bpv_ / min(tb_) / 1e9}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    a.argv = list(argv)
    if a.cpu_leg:
        return cpu_leg(a.cpu_leg)
    if a.gpus > 1 and not a.dry_run and not a.share_gpu and a.dist_backend == "nccl":
        # fail before anything touches a GPU (device_count() does not initialise HIP on this image), with one line
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print(f"bench.py: --gpus {a.gpus} asked for, {have} GPU(s) visible on this node: nothing was run", file=sys.stderr)
            sys.exit(3)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    # NEO_BENCH_FORCE_DIST=1: run the process-group path (RCCL gather on the batches' streams) even with one rank
    use_dist = world > 1 or bool(os.environ.get("NEO_BENCH_FORCE_DIST"))
    bps_default = 40
    if a.scenes is None:
        a.scenes = 256 // max(world, 8)      # cfg4: 256 scenes over the 8 GPUs of a node = 32 per GPU
    if a.config == "cfg3":
        a.batch, a.waypoints, a.no_cpu = 65536, 2, True
        a.lane_groups = True            # M = 3: eight trajectories per wavefront
        bps_default = 1
    elif a.config == "cfg4":
        a.no_cpu = True
        a.batch = 4096 * a.scenes
        bps_default = 1
    elif a.config == "cfg5":
        a.waypoints, a.grid, a.no_cpu = 40, 600, True
        bps_default = 4
    a.batches_per_step = a.batches_per_step or bps_default
    if a.dry_run:
        return dry_run(a, rank, world, 3 * a.waypoints + a.waypoints + 1)
    a.default_workload = (a.config == "cfg2" and a.batch == 4096 and a.waypoints == 20 and a.grid == 300
                          and a.dtype == "f32x" and a.layout == "brick" and not a.planar)
    # one GPU, cfg2: all three arithmetic modes are timed under the same protocol
    modes = [a.dtype] + ([m_ for m_ in ("f32x", "f32", "f64") if m_ != a.dtype]
                         if (a.config == "cfg2" and world == 1 and not a.no_modes) else [])
    R = Rank(a, rank, local_rank, world, use_dist)
    torch, dist_ = R.torch, R.dist
    B, M, D, n, n_sets = R.B, R.M, R.D, R.n, R.n_sets
    report = None
    if rank == 0 and not a.no_report:
        import bench_report as report
    esdf_build = R.esdf_build_time() if (rank == 0 and world == 1) else None

    t_gpu0 = time.time()
    main_run = R.time_mode(a.dtype)
    elapsed = main_run["elapsed"]
    # the same steps with the D2H copy of every batch's final points and statuses behind its launch (one GPU, full reports)
    host_visible = R.time_mode(a.dtype, d2h=True)["elapsed"] if (world == 1 and rank == 0 and not a.no_report) else None
    rank_rate = B * n_sets * a.steps / elapsed
    rccl_ranks, per_rank, gather_ok = None, None, None
    if use_dist:
        cdev = R.dev if a.dist_backend == "nccl" else torch.device("cpu")     # (gloo: small host tensors)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist_.all_reduce(tmax, op=dist_.ReduceOp.MAX)
        rates = torch.zeros(world, dtype=torch.float64, device=cdev)
        dist_.all_gather_into_tensor(rates, torch.tensor([rank_rate], dtype=torch.float64, device=cdev))
        per_rank = [float(v) for v in rates.cpu()]
        elapsed = float(tmax.item())
        rccl_ranks = dist_.get_world_size()
        # the gathered rows are the ranks' rows: this rank's slice bit for bit, every rank's slice by its checksum
        for bt in R.batches:
            if bt.get("work") is not None:
                bt["work"].wait()
        torch.cuda.synchronize()
        bt = R.batches[-1]
        mine = bt["gathered"][rank * B:(rank + 1) * B]
        sums = torch.zeros(world, dtype=torch.float64, device=cdev)
        dist_.all_gather_into_tensor(sums, bt["packed"].double().nan_to_num().sum().reshape(1).to(cdev))
        got = torch.stack([bt["gathered"][r_ * B:(r_ + 1) * B].double().nan_to_num().sum() for r_ in range(world)]).to(cdev)
        gather_ok = bool(torch.equal(mine.view(torch.int32), bt["packed"].view(torch.int32)) and torch.equal(got, sums))
    mode_runs = {a.dtype: main_run}
    for m_ in modes[1:]:
        mode_runs[m_] = R.time_mode(m_)
    R.bp._sync()

    if rank == 0:
        solo_ms, kernel_ms, bytes_launch = main_run["solo_ms"], main_run["kernel_ms"], main_run["bytes_launch"]
        nfev_all, nsamp_all, status_all = main_run["nfev_all"], main_run["nsamp_all"], main_run["status_all"]
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        value = world * rank_rate if not use_dist else world * B * n_sets * a.steps / elapsed
        kernel_name = ("optimize_group_kernel" if (a.lane_groups and M <= 16 and n <= 32 and a.layout != "cell8"
                                                   and a.dtype in ("f32", "f32x")) else "optimize_kernel")
        traffic = report.kernel_traffic(R, f"{kernel_name}@{B}") if report else {}
        out = {
            "metric": "trajectories/sec (batched replan)", "value": value, "unit": "traj/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "accepted_traj_per_s": value * main_run["accepted_frac"], "accepted_frac": main_run["accepted_frac"],
            # inputs resident in HBM, results copied to pinned host memory behind every launch (x: B x n doubles, status)
            "host_visible_traj_per_s": (B * n_sets * a.steps / host_visible) if host_visible else None,
            # kernel-experiment flags ORed into neo_params.flags of the timed runs (environment NEO_BENCH_FLAGS_OR); normally 0
            "flags_or": int(os.environ.get("NEO_BENCH_FLAGS_OR", "0")),
            "dispatch_order": {"pipelined_steps": "index order" if (a.no_order or a.order == "none" or (a.order == "auto" and n_sets >= R.n_lanes))
                               else "expected effort, computed on the device inside the timed region (neo_effort_order_dev)",
                               "launches_alone": "index order" if a.no_order else "expected effort, computed on the device ahead of the launch"},
            # what a caller with ONE request batch gets: a single launch of B trajectories alone on the chip (`value` keeps
            # `--streams` launches in flight and is the throughput figure)
            "single_batch_ms": solo_ms, "single_batch_traj_per_s": (B / (solo_ms * 1e-3)) if solo_ms else None,
            "single_batch_ms_range": main_run.get("solo_ms_range"),   # (mean and range over the first eight batches)
            "config": {"workload": f"{a.config}: {n_sets} x {B} trajectories x {M - 1} waypoints per step per GPU, "
                                   f"{R.n_scenes} x {a.grid}^3 {R.store} ESDF per GPU (trilinear, " +
                                   ("planar requests" if a.planar else f"pillars + {CANOPY} canopy boxes") +
                                   "), each optimised to L-BFGS-B termination"
                                   + ("; x0 = a straight line + 0.05 x the output of a RANDOM-WEIGHT initializer head (one dense "
                                      "head launch per batch in the timed region; not a trained warm start)" if R.init is not None else ""),
                       "batch_per_launch": B, "batches_per_step": n_sets, "trajectories_per_step_per_gpu": B * n_sets,
                       "pieces": M, "dims": D, "variables": n, "esdf_voxels": a.grid ** 3, "layout": a.layout,
                       "lbfgsb": "maxcor 10, maxls 20, tol 1e-4 (expert_planner.py:213-225)",
                       "sampling_arithmetic": "f32" if a.dtype == "f32x" else a.dtype,
                       "solve_and_optimiser_arithmetic": "f32" if a.dtype == "f32x" else "f64",
                       "parallelism": f"scene-sharded x{world}",
                       "launches_in_flight_per_gpu": R.n_lanes, "lane_groups": bool(a.lane_groups)},
            "rccl_ranks": rccl_ranks, "per_rank_traj_per_s": per_rank, "gather_ok": gather_ok,
            "dist_backend": (a.dist_backend if use_dist else None),
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic.get("traffic"), "traffic_source": traffic.get("source"),
                         "traffic_rule": traffic.get("rule"),
                         "kernel_ms": kernel_ms, "launches": main_run["launches"],
                         # `achieved` follows the contract: bytes of one launch / its average duration (HIP events).
                         # With several launches in flight they overlap and each one lasts longer than it
                         # would alone; the chip-wide rate is all launches' bytes over the timed region:
                         "concurrent_launches": R.n_lanes,
                         "kernel_ms_one_launch_alone": solo_ms,
                         "frac_one_launch_alone": (bytes_launch / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if solo_ms else None,
                         "achieved_aggregate": bytes_launch * n_sets * a.steps / elapsed / 1e9,
                         "frac_aggregate": bytes_launch * n_sets * a.steps / elapsed / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "evals_per_launch": float(nfev_all.sum()) / n_sets, "samples_per_launch": float(nsamp_all.sum()) / n_sets},
            "esdf_build": esdf_build,
            "optimizer": {"mean_nfev": float(nfev_all.mean()), "max_nfev": int(nfev_all.max()),
                          "mean_nit": main_run["mean_nit"],
                          "status_hist": np.bincount(status_all[0] & 0xff, minlength=7).tolist(),
                          "collision_flag_frac": float(((status_all & 0x100) != 0).mean())},
            "modes": {m_: R.mode_line(r_) for m_, r_ in mode_runs.items()},
            "headline_mode": a.dtype,
            "timed_region_s": elapsed,
            "setup_s": t_gpu0 - R.t_setup,
        }
        if R.init_report:
            out["initializer"] = R.init_report
        cpu_out = arr = None
        if R.want_cpu:
            cpu_out, arr = run_cpu_leg(a, R.g3.dist, R.res, R.synth.DOMAIN_ORIGIN, *R.sets[0])
            out["cpu_baseline"] = cpu_out["cpu_baseline"]
            out["cpu_native"] = cpu_out["cpu_native"]
        if report:
            report.extend(R, out, mode_runs, cpu_out, arr)
        details = write_details(a, out)
        # RCCL prints a version banner through C stdio; push it out first so that the JSON is the last line
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(compact_line(out, details), flush=True)
    if use_dist:
        dist_.barrier()
        dist_.destroy_process_group()


def dry_run(a, rank, world, n):
    """CPU-only exercise of the multi-rank protocol (tests/test_bench_launcher.py): process group, barrier-fenced
    timed region, per-step gather of packed results, max-over-ranks time, one JSON line from rank 0.  No device
    work happens and no throughput is claimed (value null)."""
    import torch
    import torch.distributed as dist_
    from neo_planner_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    backend = "gloo" if a.dist_backend == "nccl" else a.dist_backend
    dist_.init_process_group(backend, rank=rank, world_size=world)
    B = 64
    x = torch.full((B, n), float(rank), dtype=torch.float64)
    scenes = None
    if a.config == "cfg4":
        # BASELINE configs[3]: 256 scenes over the node, `a.scenes` = 256 // max(world, 8) per rank, rank r holding the scenes
        # r * scenes .. (r + 1) * scenes - 1 (Rank.__init__ seeds them so); every row carries its scene's number in column 1,
        # and the rank-major gather must come back scene-major
        scenes, rows = a.scenes, 2
        B = scenes * rows
        x = torch.full((B, n), float(rank), dtype=torch.float64)
        x[:, 1] = (rank * scenes + torch.arange(scenes, dtype=torch.float64)).repeat_interleave(rows)
    costs = torch.ones(B, 4, dtype=torch.float64) * (rank + 1)
    w = torch.tensor([1.0, 1.0, 1.0, 10000.0], dtype=torch.float64)
    dist_.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g = sharding.gather_results(sharding.pack_results(x, costs, w), world, force=True)
    dist_.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist_.all_reduce(el, op=dist_.ReduceOp.MAX)
    ok = all(float(g[r * B, 0]) == float(r) and abs(float(g[r * B, n]) - (r + 1) * 10003.0) < 1e-3 for r in range(world))
    extra = {}
    if scenes is not None:
        want = torch.arange(world * scenes, dtype=torch.float32).repeat_interleave(2)
        extra = {"config": "cfg4", "scenes_per_rank": scenes, "scenes_total": world * scenes,
                 "scene_major_ok": bool(torch.equal(g[:, 1], want))}
    if rank == 0:
        print(json.dumps({"metric": "trajectories/sec (batched replan)", "value": None, "unit": "traj/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * float(el) / max(a.steps, 1),
                          "dry_run": True, "ranks": dist_.get_world_size(), "gather_ok": bool(ok),
                          "dist_backend": backend, **extra}), flush=True)
    dist_.barrier()
    dist_.destroy_process_group()


if __name__ == "__main__":
    main()
