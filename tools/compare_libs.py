#!/usr/bin/env python3
"""GPU regression harness: runs a fixed set of calls through two builds of libneo_planner_hip.so and compares the
outputs BIT FOR BIT -- for refactors that must not change any arithmetic (e.g. templating the device code on the
lane-group policy).

    python tools/compare_libs.py tools/probe/old_lib/libneo_ref.so [new.so]      (default new = the in-tree build)
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
    import numpy as np
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    res = {}
    occ = synth.occupancy_3d(0, n=160, res=30.0 / 160, canopy=40)
    g3 = npa.ESDF3D.from_occupancy(occ, 30.0 / 160, synth.DOMAIN_ORIGIN)
    g16 = npa.ESDF3D.from_occupancy(occ, 30.0 / 160, synth.DOMAIN_ORIGIN, store="f16")
    m2 = npa.ESDF(); m2.occupancy_map_cb(synth.OccupancyGridMsg(synth.occupancy_2d(3)))
    for tag, M, B, lr in (("M21", 21, 512, (10.0, 28.0)), ("M3", 3, 2048, (4.0, 6.0)), ("M6", 6, 1024, (8.0, 14.0)),
                          ("M41", 41, 128, (14.0, 28.0))):
        h, t, w, ts = synth.replan_requests(5, B, M - 1, D=3, length_range=lr, **synth.VOLUME)
        for dt in ("f32", "f64"):
            bp = npa.BatchPlanner(sample_dtype=dt)
            x0 = bp.pack_x(w, ts)
            e = bp.cost_grad(g3, x0, h, t, want_coeffs=True)
            res[f"{tag}_{dt}_eval_cost"] = e["cost"]; res[f"{tag}_{dt}_eval_grad"] = e["grad"]
            s = bp.sampled_terms(g3, e["coeffs"], ts)
            res[f"{tag}_{dt}_sample_gC"] = s["grad_C"]; res[f"{tag}_{dt}_sample_gT"] = s["grad_T"]
            for wv in (1, 2):
                if dt == "f64" and wv == 2:
                    continue
                o = npa.BatchPlanner(sample_dtype=dt, waves_per_simd=wv).optimize(g3, x0, h, t)
                res[f"{tag}_{dt}_w{wv}_x"] = o["x"]; res[f"{tag}_{dt}_w{wv}_nfev"] = o["nfev"]
        if M <= 8:
            o = npa.BatchPlanner(sample_dtype="f32", lane_groups=True).optimize(g3, bp.pack_x(w, ts), h, t)
            res[f"{tag}_group_x"] = o["x"]; res[f"{tag}_group_nfev"] = o["nfev"]
            o = npa.BatchPlanner(sample_dtype="f32", lane_groups=True).optimize(g16, bp.pack_x(w, ts), h, t)
            res[f"{tag}_group16_x"] = o["x"]
    h, t, w, ts = synth.replan_requests(3, 256, 2, D=2, length_range=(4.0, 6.0))
    for dt in ("f32", "f64"):
        bp = npa.BatchPlanner(sample_dtype=dt)
        o = bp.optimize(m2, bp.pack_x(w, ts), h, t)
        res[f"map2d_{dt}_x"] = o["x"]; res[f"map2d_{dt}_nfev"] = o["nfev"]
    np.savez(out, **res)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    import numpy as np
    libs = [os.path.abspath(sys.argv[1]),
            os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(REPO, "neo-planner_amd", "neo_planner_amd", "libneo_planner_hip.so")]
    outs = []
    for i, lib in enumerate(libs):
        out = f"/tmp/neo_cmp_{i}.npz"
        env = dict(os.environ, NEO_PLANNER_LIB=lib)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out], env=env)
        outs.append(np.load(out))
    bad = 0
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        same = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
        if not same:
            bad += 1
            d = np.abs(a - b).max() / max(np.abs(a).max(), 1e-300) if a.shape == b.shape else float("nan")
            print(f"DIFF {k}: max rel {d:.3e}, {float((a != b).mean()):.3f} of entries")
    print(f"{len(outs[0].files) - bad} of {len(outs[0].files)} outputs identical bit for bit")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
