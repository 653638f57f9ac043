#!/usr/bin/env python3
"""Times the device-side 3-D ESDF construction (neo_esdf_build_3d, occupancy resident in HBM) with HIP events around
the call's kernels (neo_kernel_time of NEO_KERNEL_ESDF_BUILD) and checks it against SciPy on a smaller volume."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth, _lib

dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
for n in (int(a) for a in (sys.argv[1:] or ["300"])):
    res = 30.0 / n
    occ = synth.occupancy_3d(0, n=n, res=res, canopy=80 if n >= 300 else 0)
    d_occ = torch.from_numpy(occ).to(dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g3 = npa.ESDF3D.from_occupancy(d_occ, res, synth.DOMAIN_ORIGIN, layout=os.environ.get("NEO_LAYOUT", "brick"), ctx=ctx)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{n}^3: from_occupancy (EDT + pack, device to device) {1e3 * min(ts):.2f} ms wall (first call {1e3 * ts[0]:.1f})")
    if n <= 160:
        from scipy import ndimage
        g3 = npa.ESDF3D.from_occupancy(d_occ, res, synth.DOMAIN_ORIGIN, ctx=ctx, want_dist=True)
        want = (ndimage.distance_transform_edt(1 - occ) * res).astype(np.float32)
        print("   equal to scipy:", np.array_equal(g3.dist, want))
