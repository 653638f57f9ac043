#!/usr/bin/env python3
"""
CPU model of the ESDF-lookup kernel's cache-line traffic: which field layout and which dispatch
order make a fetched 128-byte line serve more lookups?

The model replays the lookups of `sample_kernel` for bench.py's cfg2 request batches at the initial guess:
  * samples per piece and the balanced lane assignment of csrc/neo_device.hpp (balanced_sample_lanes): round `it` of a
    wavefront touches samples [it * Lp, (it + 1) * Lp) of every piece p;
  * workgroup b runs on XCD b mod 8 (MI355X_MICROARCH.md: round-robin dispatch), each XCD has a 4 MB L2 (LRU here,
    32768 lines); the wavefronts resident on an XCD advance round by round, interleaved;
  * a lookup touches the line(s) holding its 8 corners in the given layout.
Output per (layout, order): lines fetched from the fabric, lookups per fetched line, L2 hit rate, traffic / algorithmic.
It is a model (no L1, no set conflicts, idealised interleaving): it ranks variants before GPU time is spent on them;
the counters under profiles/ are the measurement.

    python tools/sim_esdf_locality.py [--batches 1] [--grid 300]
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))

from neo_planner_amd import synth  # noqa: E402
from neo_planner_amd import sharding  # noqa: E402,F401


def coefficients(head, tail, wp, ts):
    """polynomial coefficients of the initial guesses [B][M][6][D] (cpu_native: the reference's banded solve)"""
    from oracle import cpu_native as cn
    from oracle import minco_np as onp
    B, M = ts.shape
    D = head.shape[2]
    dist = np.full((4, 4, 4), 5.0, np.float32)
    nm = cn.NativeMap.from_field3d(dist, 10.0, (0.0, -20.0, 0.0))
    cfg = onp.PlannerParams()
    out = np.zeros((B, M, 6, D))
    pl = cn.NativePlanner(cfg)
    for b in range(B):
        pl.read_planning_conditions(nm, head[b], tail[b], wp[b], ts[b])
        x = np.concatenate([wp[b].reshape(-1), pl.map_T2tau(ts[b])])
        pl.get_grad(x)
        out[b] = pl.coeffs.reshape(M, 6, D)
    return out


def lookups(coeffs, ts, grid, res, origin, dt=0.1):
    """per trajectory: list over rounds of int arrays [k][3] of cell indices (ix, iy, iz) touched in that round"""
    B, M = ts.shape
    ns = np.floor(ts / dt).astype(int)
    per_traj = []
    for b in range(B):
        total = int(ns[b].sum())
        R = max(1, -(-total // 64))
        while True:
            Lp = -(-ns[b] // R)
            if Lp.sum() <= 64:
                break
            R += 1
        rounds = [[] for _ in range(R)]
        for p in range(M):
            j = np.arange(ns[b, p])
            t = j * dt
            pw = np.stack([t ** k for k in range(6)], axis=1)          # [J][6]
            pos = pw @ coeffs[b, p]                                     # [J][3]
            u = (pos - np.asarray(origin)) / res - 0.5
            inside = ((u >= -0.5) & (u < grid - 0.5)).all(axis=1)
            i0 = np.clip(np.floor(u).astype(np.int64), 0, grid - 2)
            i0[~inside] = 0
            it = j // Lp[p]
            for r in range(R):
                sel = it == r
                if sel.any():
                    rounds[r].append(i0[sel])
        per_traj.append([np.concatenate(r) if r else np.zeros((0, 3), np.int64) for r in rounds])
    return per_traj


# ---- layouts: cell (ix, iy, iz) -> array of line ids it touches (one row per lookup, -1 = unused slot)
def lines_yz4(c, grid, esz=4):
    rec = 4 * esz                                    # bytes per voxel record
    v = (c[:, 2] * grid + c[:, 1]) * grid + c[:, 0]
    a0 = v * rec
    a1 = v * rec + 2 * rec - 1
    return np.stack([a0 // 128, np.where(a1 // 128 != a0 // 128, a1 // 128, -1)], axis=1)


def lines_linear(c, grid, esz=4):
    out = []
    for dz in (0, 1):
        for dy in (0, 1):
            v = ((c[:, 2] + dz) * grid + c[:, 1] + dy) * grid + c[:, 0]
            out.append(v * esz // 128)
            out.append((v * esz + 2 * esz - 1) // 128)
    return np.stack(out, axis=1)


def lines_brick(c, grid, bx, by, bz):
    """one line per block of bx x by x bz CELLS (the block's (bx+1)(by+1)(bz+1) corners stored together)"""
    nbx, nby = -(-grid // bx), -(-grid // by)
    return (((c[:, 2] // bz) * nby + c[:, 1] // by) * nbx + c[:, 0] // bx)[:, None]


def lines_yz4_tiled(c, grid, ty, tz, esz=4):
    """yz-quad records, but lines of 8 x-records ordered so that a (ty x tz) tile of (y, z) rows is contiguous: the line
    a lookup needs is the same as in yz4 -- only neighbouring lines sit in the same DRAM page / L2 set; identical line
    count, listed to make that explicit"""
    return lines_yz4(c, grid, esz)


LAYOUTS = {
    "yz4": lambda c, g: lines_yz4(c, g),
    "linear": lambda c, g: lines_linear(c, g),
    "brick2x2x2 (27 corners, 108 B)": lambda c, g: lines_brick(c, g, 2, 2, 2),
    "brick4x2x1 (30 corners, 120 B)": lambda c, g: lines_brick(c, g, 4, 2, 1),
    "brick4x1x2": lambda c, g: lines_brick(c, g, 4, 1, 2),
    "brick3x2x1 (24 corners)": lambda c, g: lines_brick(c, g, 3, 2, 1),
    "brick7x1x1 (=yz4 w/o straddle)": lambda c, g: lines_brick(c, g, 7, 1, 1),
    "brick3x1x1 x2 lines 256B: 3x3x... n/a": None,
}


def morton3(x, y, z, bits=5):
    k = np.zeros_like(x)
    for b in range(bits):
        k |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return k


def orders(head, tail, B):
    """dispatch orders: position i of the returned permutation = trajectory run by workgroup i"""
    out = {"index": np.arange(B)}
    mid = 0.5 * (head[:, 0] + tail[:, 0])
    # coarse cells of 1 m on the start (y, z) -- all requests start near x = 0..3 and fly towards +x -- then the heading
    sy = np.floor((head[:, 0, 1] + 15.0) / 1.0).astype(np.int64)
    sz = np.floor(head[:, 0, 2] / 1.0).astype(np.int64)
    ey = np.floor((tail[:, 0, 1] + 15.0) / 2.0).astype(np.int64)
    ez = np.floor(tail[:, 0, 2] / 2.0).astype(np.int64)
    key = morton3(sy, sz, np.zeros_like(sy), bits=5) * 4096 + morton3(ey, ez, np.zeros_like(ey), bits=4)
    srt = np.argsort(key, kind="stable")
    # XCD-aware deal: workgroup i runs on XCD i mod 8; give XCD k the k-th contiguous eighth of the sorted list
    per = B // 8
    deal = np.empty(B, dtype=np.int64)
    for k in range(8):
        deal[k::8] = srt[k * per:(k + 1) * per]
    out["sorted, dealt by XCD"] = deal
    out["sorted only (no XCD deal)"] = srt
    km = morton3(np.floor(mid[:, 1] + 15).astype(np.int64), np.floor(mid[:, 2]).astype(np.int64), np.zeros(B, np.int64))
    sm = np.argsort(km, kind="stable")
    d2 = np.empty(B, dtype=np.int64)
    for k in range(8):
        d2[k::8] = sm[k * per:(k + 1) * per]
    out["sorted by midpoint (y,z), dealt by XCD"] = d2
    return out


def simulate(per_traj, order, line_fn, grid, resident_per_xcd=512, l2_lines=32768):
    """returns (lookups, line requests, fabric fetches)"""
    B = len(order)
    n_lookups = n_req = n_fetch = 0
    for xcd in range(8):
        mine = order[xcd::8]
        cache = OrderedDict()
        for w0 in range(0, len(mine), resident_per_xcd):
            group = [per_traj[b] for b in mine[w0:w0 + resident_per_xcd]]
            for r in range(max(len(t) for t in group)):
                for t in group:
                    if r >= len(t) or len(t[r]) == 0:
                        continue
                    ln = line_fn(t[r], grid)
                    n_lookups += len(t[r])
                    u = np.unique(ln[ln >= 0])          # one request per distinct line of the wave instruction
                    n_req += len(u)
                    for q in u.tolist():
                        if q in cache:
                            cache.move_to_end(q)
                        else:
                            n_fetch += 1
                            cache[q] = True
                            if len(cache) > l2_lines:
                                cache.popitem(last=False)
    return n_lookups, n_req, n_fetch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=1)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--layouts", default=None, help="comma-separated prefixes of the layouts to run (default: all)")
    a = ap.parse_args()
    res = 30.0 / a.grid
    parts = [synth.replan_requests(1000 * r, a.batch, 20, D=3, **synth.VOLUME) for r in range(a.batches)]
    head, tail, wp, ts = (np.concatenate([p[k] for p in parts]) for k in range(4))
    B = len(ts)
    cf = coefficients(head, tail, wp, ts)
    pt = lookups(cf, ts, a.grid, res, synth.DOMAIN_ORIGIN)
    n_samples = sum(len(r) for t in pt for r in t)
    alg = n_samples * 32 + B * (2 * 81 * 4 + 20)
    print(f"{B} trajectories, {n_samples} lookups, algorithmic bytes {alg / 1e6:.1f} MB")
    ords = orders(head, tail, B)
    for lname, fn in LAYOUTS.items():
        if fn is None or (a.layouts and not any(lname.startswith(q) for q in a.layouts.split(","))):
            continue
        allc = np.concatenate([r for t in pt for r in t])
        ln = fn(allc, a.grid)
        distinct = len(np.unique(ln[ln >= 0]))
        within = sum(len(np.unique(fn(np.concatenate(t), a.grid))) - (1 if (fn(np.concatenate(t), a.grid) < 0).any() else 0)
                     for t in pt)
        print(f"\n== layout {lname}: distinct lines {distinct} ({distinct * 128 / 1e6:.1f} MB), per-trajectory distinct "
              f"{within} ({n_samples / within:.2f} lookups/line with a perfect per-wavefront cache)")
        for oname, od in ords.items():
            nl, nr, nf = simulate(pt, od, fn, a.grid)
            print(f"   order {oname:42s}: fetched {nf:8d} lines = {nf * 128 / 1e6:7.1f} MB, lookups/fetched line {nl / nf:5.2f}, "
                  f"L2 hit {1 - nf / nr:5.3f}, traffic/algorithmic {nf * 128 / alg:5.2f}")


if __name__ == "__main__":
    main()
