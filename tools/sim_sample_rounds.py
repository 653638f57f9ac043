#!/usr/bin/env python3
"""
CPU model: rounds of the sample loop (minco_sample, csrc/neo_device.hpp) along real optimisation runs of cfg2 requests
under three lane assignments:
  current   the smallest R with sum_p ceil(ns_p / R) <= 64 (balanced_sample_lanes)
  merged    floor(ns_p / R) full lanes a piece, the remainders (< R samples each) packed two to a lane
  ideal     ceil(sum ns_p / 64)
The runs are the pinned C++ oracle's (test infrastructure; this is an analysis script, not a product path).

    python tools/sim_sample_rounds.py [--requests 48] [--grid 150]
"""
import argparse, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
from neo_planner_amd import synth
from oracle import cpu_native as cn
from oracle import minco_np as onp


def rounds_current(ns):
    R = max(1, -(-int(ns.sum()) // 64))
    while np.sum(-(-ns // R)) > 64:
        R += 1
    return R


def rounds_merged(ns):
    R = max(1, -(-int(ns.sum()) // 64))
    while True:
        full = int(np.sum(ns // R))
        rem = np.sort(ns % R)[::-1]
        rem = rem[rem > 0]
        # pairs: largest with the smallest that still fits (two remainders a lane at most)
        lanes, lo, hi = 0, 0, len(rem) - 1
        while lo <= hi:
            if lo < hi and rem[lo] + rem[hi] <= R:
                hi -= 1
            lo += 1
            lanes += 1
        if full + lanes <= 64:
            return R
        R += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=48)
    ap.add_argument("--grid", type=int, default=150)
    a = ap.parse_args()
    from scipy import ndimage
    res = 30.0 / a.grid
    occ = synth.occupancy_3d(0, n=a.grid, res=res, canopy=40 if a.grid >= 150 else 0)
    dist = (ndimage.distance_transform_edt(1 - occ) * res).astype(np.float32)
    nm = cn.NativeMap.from_field3d(dist, res, synth.DOMAIN_ORIGIN)
    head, tail, wp, ts = synth.replan_requests(0, a.requests, 20, D=3, **synth.VOLUME)
    cfg = onp.PlannerParams()
    tot = dict(current=0, merged=0, ideal=0, evals=0, samples=0)
    first = dict(current=0, merged=0, ideal=0)
    for b in range(a.requests):
        pl = cn.NativePlanner(cfg)
        pl.read_planning_conditions(nm, head[b], tail[b], wp[b], ts[b])
        tr = []
        try:
            pl.plan_once(trace=tr)
        except (ValueError, OverflowError):
            pass
        xs = [np.concatenate([wp[b].reshape(-1), pl.map_T2tau(ts[b])])] + [t[0] for t in tr]
        for k, x in enumerate(xs):
            T = pl.map_tau2T(x[-21:])
            ns = np.floor(T / cfg.delta_t).astype(np.int64)
            c, m, i = rounds_current(ns), rounds_merged(ns), max(1, -(-int(ns.sum()) // 64))
            tot["current"] += c; tot["merged"] += m; tot["ideal"] += i; tot["evals"] += 1; tot["samples"] += int(ns.sum())
            if k == 0:
                first["current"] += c; first["merged"] += m; first["ideal"] += i
    e = tot["evals"]
    print(f"{a.requests} runs, {e} iterates: samples per iterate {tot['samples'] / e:.0f}")
    print(f"rounds per iterate: current {tot['current'] / e:.2f}  merged tails {tot['merged'] / e:.2f}  ideal {tot['ideal'] / e:.2f}")
    print(f"at the initial guess: current {first['current'] / a.requests:.2f}  merged {first['merged'] / a.requests:.2f}  ideal {first['ideal'] / a.requests:.2f}")


if __name__ == "__main__":
    main()
