#!/usr/bin/env python3
"""Two builds of the library on an fp16 field (cfg5's arithmetic: M = 41, fp16 brick field): the product's blend -- fp16
storage, every operation in fp32 -- against a `-DNEO_F16_PACKED_BLEND` build (y / z interpolation in packed fp16).

    python tools/experiments/gpu_f16_blend.py <packed.so> [grid]

Per evaluation: cost and gradient of both builds against the product's fp64-arithmetic evaluation on the SAME fp16 field
(relative errors); whole runs: evaluations, exits, accepted share, final costs."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(out, grid):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
    import numpy as np
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    res = 30.0 / grid
    occ = synth.occupancy_3d(0, n=grid, res=res, canopy=80)
    g16 = npa.ESDF3D.from_occupancy(occ, res, synth.DOMAIN_ORIGIN, store="f16", layout="brick")
    M, B = 41, 2048
    h, t, w, ts = synth.replan_requests(5, B, M - 1, D=3, **synth.VOLUME)
    r = {}
    bp64 = npa.BatchPlanner(sample_dtype="f64")
    x0 = bp64.pack_x(w, ts)
    e64 = bp64.cost_grad(g16, x0, h, t)
    r["cost64"], r["grad64"] = e64["cost"], e64["grad"]
    bpx = npa.BatchPlanner(sample_dtype="f32x")
    e = bpx.cost_grad(g16, x0, h, t)
    r["cost"], r["grad"] = e["cost"], e["grad"]
    o = bpx.optimize(g16, x0, h, t)
    r["nfev"], r["status"], r["costs"], r["collision"] = o["nfev"], o["status"], o["costs"], o["collision"]
    # a second evaluation point: the optimised x (samples near obstacles: the collision term is active there)
    e2 = bp64.cost_grad(g16, o["x"], h, t)
    e3 = bpx.cost_grad(g16, o["x"], h, t)
    r["cost64_opt"], r["grad64_opt"], r["cost_opt"], r["grad_opt"] = e2["cost"], e2["grad"], e3["cost"], e3["grad"]
    np.savez(out, **r)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    import numpy as np
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    libs = {"product": os.path.join(REPO, "neo-planner_amd", "neo_planner_amd", "libneo_planner_hip.so"), "packed": os.path.abspath(sys.argv[1])}
    for tag, lib in libs.items():
        out = f"/tmp/neo_f16_{tag}.npz"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, str(grid)], env=dict(os.environ, NEO_PLANNER_LIB=lib))
        d = np.load(out)
        for sfx, name in (("", "start points"), ("_opt", "optimised points")):
            c, c64, g, g64 = d["cost" + sfx], d["cost64" + sfx], d["grad" + sfx], d["grad64" + sfx]
            ok = np.isfinite(c) & np.isfinite(c64)
            rc = np.abs(c - c64)[ok] / np.maximum(np.abs(c64[ok]), 1e-30)
            rg = np.linalg.norm(g - g64, axis=1)[ok] / np.maximum(np.linalg.norm(g64, axis=1)[ok], 1e-30)
            print(f"{tag:8s} {name:17s} cost rel err median {np.median(rc):.2e} p99 {np.quantile(rc, 0.99):.2e} max {rc.max():.2e} | "
                  f"gradient rel err median {np.median(rg):.2e} p99 {np.quantile(rg, 0.99):.2e} max {rg.max():.2e}")
        st = d["status"] & 0xff
        acc = ((st <= 2) & ~d["collision"]).mean()
        print(f"{tag:8s} runs: mean nfev {d['nfev'].mean():.1f} max {d['nfev'].max()} accepted {acc:.4f} exits {np.bincount(st, minlength=7).tolist()} "
              f"median collision term {np.median(d['costs'][:, 3]):.3e}")


if __name__ == "__main__":
    main()
