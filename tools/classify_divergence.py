#!/usr/bin/env python3
"""First-divergence classification of optimiser runs that end differently.

Two runs of the same request are laid side by side evaluation by evaluation -- records (f, line-search step,
quadrature samples, iteration) from `neo_optimize_trace` on the GPU and from oracle/cpu_native on the host -- and the
FIRST evaluation at which they differ is classified:

  sample_count_jump        same iteration, same step (= the same trial point), but sum_i int(T_i / dt) differs: a
                           duration sits on a multiple of dt and the two evaluations count one sample apart
  objective_jump           same trial point, same sample count, f differs by more than 1e-9 relative
  line_search_step         same iteration, different step: dcsrch/dcstep picked another trial step from inputs that
                           agreed to round-off at the previous evaluation (a comparison decided by the last bits)
  line_search_termination  different iteration number: one run accepted the step (or restarted) where the other
                           kept searching
  termination              all common evaluations agree, one run stopped earlier (convergence tests on round-off)

Besides the first-divergence classes the summary carries `growth`: the median (and 90th percentile) relative difference
of f between the two runs at evaluation k = 1, 5, 10, 20, 30, 50, 80, 120, 200 over the runs still going, and the share
of runs whose sample counts differ there.  On this objective the difference grows roughly tenfold every 20 evaluations
from round-off level -- the optimiser's path is sensitive to the last bits long before any discrete event (a sample-count
jump) occurs; the discrete classes mostly mark where that smooth growth crosses `rtol`.

Modes:
  --cpu-only        cpu_native against itself with the coefficients perturbed by one ulp (runs anywhere)
  (default, GPU)    the HIP path with fp64 sampling against cpu_native, batch 0 of bench.py's cfg2 workload
Writes a JSON summary (stdout and --out)."""
import argparse
import ctypes
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np


def classify(tr_a, nfev_a, tr_b, nfev_b, rtol=1e-9):
    """per trajectory: (category, first differing evaluation index or -1)"""
    out = []
    for a, na, b, nb in zip(tr_a, nfev_a, tr_b, nfev_b):
        k_end = int(min(na, nb, a.shape[0], b.shape[0]))
        cat, where = "identical", -1
        for k in range(k_end):
            fa, sa, nsa, ita = a[k]
            fb, sb, nsb, itb = b[k]
            same_f = abs(fa - fb) <= rtol * max(abs(fa), abs(fb), 1e-300)
            same_s = abs(sa - sb) <= rtol * max(abs(sa), abs(sb), 1e-300)
            if ita == itb and same_s and nsa == nsb and same_f:
                continue
            where = k
            if ita != itb:
                cat = "line_search_termination"
            elif not same_s:
                cat = "line_search_step"
            elif nsa != nsb:
                cat = "sample_count_jump"
            else:
                cat = "objective_jump"
            break
        if cat == "identical" and na != nb:
            cat, where = "termination", k_end
        out.append((cat, where))
    return out


def growth(tr_a, nfev_a, tr_b, nfev_b, ks=(1, 5, 10, 20, 30, 50, 80, 120, 200)):
    rows = []
    for k in ks:
        ok = (np.asarray(nfev_a) > k) & (np.asarray(nfev_b) > k)
        if k >= tr_a.shape[1] or ok.sum() < 8:
            break
        fa, fb = tr_a[ok, k, 0], tr_b[ok, k, 0]
        rel = np.abs(fa - fb) / np.maximum(np.abs(fa), 1e-300)
        rows.append(dict(evaluation=k, runs=int(ok.sum()), median_rel_df=float(np.median(rel)),
                         p90_rel_df=float(np.quantile(rel, 0.9)),
                         frac_sample_count_differs=float((tr_a[ok, k, 2] != tr_b[ok, k, 2]).mean())))
    return rows


def summarise(cls, nfev_a, nfev_b):
    cats = {}
    for c, _ in cls:
        cats[c] = cats.get(c, 0) + 1
    n = len(cls)
    differing = [(c, w) for (c, w), a, b in zip(cls, nfev_a, nfev_b) if c != "identical"]
    return dict(n=n, frac_same_nfev=float(np.mean(np.asarray(nfev_a) == np.asarray(nfev_b))),
                first_divergence={k: v / n for k, v in sorted(cats.items())},
                median_first_differing_evaluation=float(np.median([w for _, w in differing])) if differing else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--cap", type=int, default=600)
    ap.add_argument("--cpu-only", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from neo_planner_amd import synth
    from oracle import cpu_native as cn
    M, D, N = 21, 3, a.n
    head, tail, wp, ts = synth.replan_requests(0, 4096, M - 1, D=D, **synth.VOLUME)
    head, tail, wp, ts = head[:N], tail[:N], wp[:N], ts[:N]
    tau = -np.log((5.0 - 0.5) / (ts - 0.5) - 1.0)
    x0 = np.concatenate([wp.reshape(N, -1), tau], axis=1)
    threads = len(os.sched_getaffinity(0))
    res = {}
    if a.cpu_only:
        dist = synth.esdf_3d(0, canopy=80)
    else:
        import torch
        import neo_planner_amd as npa
        dev = torch.device("cuda", 0)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        ctx = npa.Context(0, stream=st.cuda_stream)
        occ = synth.occupancy_3d(0, canopy=80)
        g3 = npa.ESDF3D.from_occupancy(torch.from_numpy(occ).to(dev), 0.1, synth.DOMAIN_ORIGIN, ctx=ctx, want_dist=True)
        dist = g3.dist
        for mode in ("f64", "f32"):
            bp = npa.BatchPlanner(ctx=ctx, sample_dtype=mode, waves_per_simd=1)
            bp._sync()
            x = torch.from_numpy(x0).to(dev)
            tr = torch.zeros(N, a.cap, 4, dtype=torch.float64, device=dev)
            ctx.check(ctx.lib.neo_optimize_trace(ctx.h, ctypes.c_void_p(tr.data_ptr()), a.cap))
            costs = torch.zeros(N, 4, dtype=torch.float64, device=dev); last = torch.zeros_like(costs)
            nit = torch.zeros(N, dtype=torch.int32, device=dev); nfev = torch.zeros_like(nit); stt = torch.zeros_like(nit)
            bp.optimize_dev(g3, x, torch.from_numpy(head).to(dev), torch.from_numpy(tail).to(dev), costs, last, nit, nfev, stt)
            torch.cuda.synchronize()
            ctx.check(ctx.lib.neo_optimize_trace(ctx.h, None, 0))
            res["gpu_" + mode] = (tr.cpu().numpy(), nfev.cpu().numpy())
    nm = cn.NativeMap.from_field3d(dist, 0.1, synth.DOMAIN_ORIGIN)
    base = cn.optimize_batch(nm, x0, head, tail, M, D, threads=threads, trace_cap=a.cap)
    out = {"workload": f"first {N} requests of bench.py's cfg2 batch 0 (scene 0, volume requests)", "rtol": 1e-9}
    pert = cn.optimize_batch(nm, x0, head, tail, M, D, params=cn.make_params(coeff_eps=2.2e-16), threads=threads,
                             trace_cap=a.cap)
    out["control_cpu_vs_cpu_coeffs_1ulp"] = summarise(classify(base["trace"], base["nfev"], pert["trace"], pert["nfev"]),
                                                      base["nfev"], pert["nfev"])
    out["control_cpu_vs_cpu_coeffs_1ulp"]["growth"] = growth(base["trace"], base["nfev"], pert["trace"], pert["nfev"])
    p32 = cn.optimize_batch(nm, x0, head, tail, M, D, params=cn.make_params(sample_f32=True), threads=threads,
                            trace_cap=a.cap)
    out["control_cpu_vs_cpu_fp32_sampling"] = summarise(classify(base["trace"], base["nfev"], p32["trace"], p32["nfev"], 1e-4),
                                                        base["nfev"], p32["nfev"])
    out["control_cpu_vs_cpu_fp32_sampling"]["rtol"] = 1e-4
    out["control_cpu_vs_cpu_fp32_sampling"]["growth"] = growth(base["trace"], base["nfev"], p32["trace"], p32["nfev"])
    for k, (tr, nf) in res.items():
        rt = 1e-9 if k == "gpu_f64" else 1e-4
        out[k + "_vs_cpu_native"] = summarise(classify(tr, nf, base["trace"], base["nfev"], rt), nf, base["nfev"])
        out[k + "_vs_cpu_native"]["rtol"] = rt
        out[k + "_vs_cpu_native"]["growth"] = growth(tr, nf, base["trace"], base["nfev"])
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
