#!/usr/bin/env python3
"""eval_kernel launches with phases switched off (neo_params.flags debug bits), for a rocprofv3 --pmc pass:
instruction counts per phase = differences between the dispatches (in launch order)."""
import ctypes, os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth
grid = 300; res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res)
B, M, D = int(os.environ.get("NEO_B", "4096")), int(os.environ.get("NEO_M", "21")), 3
head, tail, wp, ts = synth.replan_requests(0, B, M - 1, D=D, length_range=((4.0, 6.0) if M == 3 else (10.0, 28.0)))
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = npa.Context(0, stream=st.cuda_stream)
g3 = npa.ESDF3D(torch.from_numpy(dist).to(dev), res, synth.DOMAIN_ORIGIN, store="f32", ctx=ctx)
n = D * (M - 1) + M
bp = npa.BatchPlanner(ctx=ctx, sample_dtype="f32")
x = torch.from_numpy(bp.pack_x(wp, ts)).to(dev)
h = torch.from_numpy(head).to(dev); tl = torch.from_numpy(tail).to(dev)
cost = torch.zeros(B, dtype=torch.float64, device=dev); c4 = torch.zeros(B, 4, dtype=torch.float64, device=dev)
g = torch.zeros(B, n, dtype=torch.float64, device=dev); stt = torch.zeros(B, dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for dbg in (0, 1, 2, 3, 8, 16):   # full, no sample loop, no sweeps, neither, no factor, no scans
    bp._sync(); ctx.set_params(flags=dbg)
    ctx.check(ctx.lib.neo_cost_grad_batch_dev(ctx.h, g3.scene_id, B, M, D, p(x), p(h), p(tl), p(cost), p(c4), p(g), None, p(stt)))
    torch.cuda.synchronize()
print("done")
