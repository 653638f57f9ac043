import sys, os, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "neo-planner_amd"))
import numpy as np, torch
import neo_planner_amd as npa
from neo_planner_amd import synth
grid=300; res=30.0/grid
dist=synth.esdf_3d(0,n=grid,res=res)
dev=torch.device("cuda",0)
st=torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx=npa.Context(0,stream=st.cuda_stream)
g3=npa.ESDF3D(torch.from_numpy(dist).to(dev),res,synth.DOMAIN_ORIGIN,store="f32",ctx=ctx)
M,D=21,3
for B in (2048,4096,8192,16384):
    head,tail,wp,ts=synth.replan_requests(0,B,M-1,D=D)
    for waves in (1,2):
        bp=npa.BatchPlanner(ctx=ctx,sample_dtype="f32",waves_per_simd=waves); bp._sync()
        x0=torch.from_numpy(bp.pack_x(wp,ts)).to(dev); x=x0.clone()
        h=torch.from_numpy(head).to(dev); tl=torch.from_numpy(tail).to(dev)
        costs=torch.zeros(B,4,dtype=torch.float64,device=dev); last=torch.zeros_like(costs)
        nit=torch.zeros(B,dtype=torch.int32,device=dev); nfev=torch.zeros_like(nit); status=torch.zeros_like(nit)
        order=torch.from_numpy(bp.expected_effort_order(head,tail,ts).astype(np.int32)).to(dev)
        ctx.check(ctx.lib.neo_optimize_dispatch_order(ctx.h,ctypes.c_void_p(order.data_ptr()),B))
        def run():
            x.copy_(x0); bp.optimize_dev(g3,x,h,tl,costs,last,nit,nfev,status)
        run(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(6): run()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/6
        print(f"B={B} waves={waves}: {dt*1e3:.2f} ms  {B/dt:.0f} traj/s")
