#!/usr/bin/env python3
"""Random volumes through neo_esdf_build_3d against scipy.ndimage.distance_transform_edt (equal after the fp32 rounding):
shapes with x rows of every alignment, lines around powers of two, sparse / dense / empty rows, both forms of the line
passes.   python tools/experiments/gpu_edt_fuzz.py [cases] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
from scipy import ndimage
import neo_planner_amd as npa

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    pick = lambda: int(rng.choice([rng.integers(2, 12), rng.integers(12, 70), rng.choice([4, 8, 16, 32, 64, 128, 3, 5, 17, 33, 65, 129, 130, 257])]))
    shape = (pick(), pick(), pick())
    if shape[0] * shape[1] * shape[2] > 4_000_000:
        continue
    dens = float(rng.choice([0.0005, 0.005, 0.05, 0.4]))
    occ = (rng.random(shape) < dens).astype(np.uint8)
    if rng.random() < 0.5:
        occ[0] = 1
    if rng.random() < 0.3:
        occ[:, : shape[1] // 2, :] = 0
    if occ.sum() == 0:
        occ[tuple(rng.integers(0, s) for s in shape)] = 1
    res = float(rng.choice([0.1, 0.25, 0.07]))
    want = (ndimage.distance_transform_edt(1 - occ) * res).astype(np.float32)
    for generic in (0, 1):
        ctx_ = npa._lib.default_context()
        ctx_.check(ctx_.lib.neo_esdf_build_config(ctx_.h, generic))     # 1 = NEO_EDT_GENERIC_LINES
        layout = str(rng.choice(["linear", "brick", "yz4"]))
        g3 = npa.ESDF3D.from_occupancy(occ, res, (0.0, 0.0, 0.0), layout=layout, want_dist=True)
        if not np.array_equal(g3.dist, want):
            bad += 1
            print("MISMATCH", shape, dens, generic, layout, int((g3.dist != want).sum()))
print(f"{cases} cases, {bad} mismatches")
