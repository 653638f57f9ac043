#!/usr/bin/env python3
"""y-pass duration of the 3-D EDT against the number of blocks (volumes 300 x 300 x nz): does the time grow with the grid
once every CU has one block, or only once it has five (the LDS limit)?  Run under rocprofv3 --kernel-trace --stats."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
for nz in (int(a) for a in (sys.argv[1:] or ["14", "28", "70", "140", "300"])):
    occ = (rng.random((nz, 300, 300)) < 0.002).astype(np.uint8)
    occ[0] = 1
    d = torch.from_numpy(occ).cuda()
    for _ in range(3):
        g = npa.ESDF3D.from_occupancy(d, 0.1, (0.0, 0.0, 0.0), layout="linear", ctx=ctx)
    torch.cuda.synchronize()
