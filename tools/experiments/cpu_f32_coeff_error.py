#!/usr/bin/env python3
"""Round 6: where the all-fp32 mode's coefficient error comes from.  The device formulation of get_coeffs
(expert_planner.py:261-336 as the block-tridiagonal joint system + the Hermite form of each quintic, neo_device.hpp
minco_forward) in NumPy with each stage in fp32 or fp64, on 64 cfg2 requests with durations scaled by U[0.5, 1.3]; error
per coefficient order k, relative to max |c_k| of the trajectory, against the all-fp64 pipeline.  CPU only.

    python tools/experiments/cpu_f32_coeff_error.py > profiles/r06_f32_coeff_error.txt

Reading: durations rounded to fp32 ALONE -- everything else fp64 -- already move c3..c5 by 0.7 - 1.3e-6 (max 3.7e-6),
waypoints rounded to fp32 alone by 0.6 - 0.8e-6; the fp32 solve leaves 8e-7 on (v, a), which the Hermite form carries to
2 - 4e-6 on c3..c5 whatever precision the Hermite form itself runs in.  One step of iterative refinement removes only the
last part: 2e-7 on the coefficients needs fp64 durations, displacements, residuals and Hermite differences -- that is the
`f32` mode (fp64 solve, fp32 sampling)."""
import sys, numpy as np
import os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'neo-planner_amd'))
from neo_planner_amd import synth
from oracle import minco_np as onp
f32=np.float32
def pipeline(wp, ts, head, tail, dt_solve, dt_herm, dt_T=None, dt_rhs=None, compensated=False):
    """block-tridiagonal formulation of neo_device.hpp in chosen precisions; returns coeffs [M,6,D] float64"""
    dt_T = dt_T or dt_solve; dt_rhs = dt_rhs or dt_solve
    M=len(ts); D=wp.shape[0]
    T=np.asarray(ts,dt_T)
    P=np.concatenate([head[0][:,None], wp, tail[0][:,None]],1).T  # [M+1,D]
    P=P.astype(dt_rhs)
    i1=(1/T).astype(dt_solve); i2=i1*i1; i3=i2*i1; i4=i2*i2
    n=M-1
    K=np.zeros((2*n,2*n),dt_solve); R=np.zeros((2*n,D),dt_solve)
    dP=(P[1:]-P[:-1]).astype(dt_solve)
    for j in range(1,M):  # joint j between piece j-1 (a) and j (b)
        a1,a2,a3,a4=i1[j-1],i2[j-1],i3[j-1],i4[j-1]; b1,b2,b3,b4=i1[j],i2[j],i3[j],i4[j]
        Lo=np.array([[-24*a2,-3*a1],[-168*a3,-24*a2]],dt_solve)
        Di=np.array([[-36*a2+36*b2, 9*a1+9*b1],[-192*a3-192*b3, 36*a2-36*b2]],dt_solve)
        Up=np.array([[24*b2,-3*b1],[-168*b3,24*b2]],dt_solve)
        r=np.stack([-(60*a3*dP[j-1]-60*b3*dP[j]), -(360*a4*dP[j-1]+360*b4*dP[j])]).astype(dt_solve)
        k=j-1
        K[2*k:2*k+2,2*k:2*k+2]=Di
        if k>0: K[2*k:2*k+2,2*k-2:2*k]=Lo
        else: r-= Lo@np.stack([head[1],head[2]]).astype(dt_solve)
        if k<n-1: K[2*k:2*k+2,2*k+2:2*k+4]=Up
        else: r-= Up@np.stack([tail[1],tail[2]]).astype(dt_solve)
        R[2*k:2*k+2]=r
    z=np.linalg.solve(K,R).astype(dt_solve)
    V=np.concatenate([head[1][None],z[0::2],tail[1][None]]).astype(dt_herm)
    A=np.concatenate([head[2][None],z[1::2],tail[2][None]]).astype(dt_herm)
    Th=T.astype(dt_herm)[:,None]; Ph=P.astype(dt_herm)
    T2=Th*Th
    ep=Ph[1:]-Ph[:-1]-Th*V[:-1]-dt_herm(0.5)*T2*A[:-1]
    ev=V[1:]-V[:-1]-Th*A[:-1]
    ea=A[1:]-A[:-1]
    j1=(1/Th).astype(dt_herm); j3=j1*j1*j1; j4=j3*j1
    c=np.zeros((M,6,D))
    c[:,0]=Ph[:-1]; c[:,1]=V[:-1]; c[:,2]=dt_herm(0.5)*A[:-1]
    c[:,3]=(10*ep-4*Th*ev+dt_herm(0.5)*T2*ea)*j3
    c[:,4]=(-15*ep+7*Th*ev-T2*ea)*j4
    c[:,5]=(6*ep-3*Th*ev+dt_herm(0.5)*T2*ea)*j4*j1
    return c
M,D=21,3
head,tail,wp,ts=synth.replan_requests(0,64,M-1,D=D,**synth.VOLUME)
def pad(a): 
    o=np.zeros((3,a.shape[1])); o[:a.shape[0]]=a; return o
res={}
for b in range(64):
    h=pad(head[b]); t=pad(tail[b])
    # jitter durations like an optimiser would
    rng=np.random.default_rng(b); tsb=ts[b]*rng.uniform(0.5,1.3,M)
    ref=pipeline(wp[b],tsb,h,t,np.float64,np.float64)
    for name,args in dict(all32=(f32,f32), solve64=(np.float64,f32), herm64=(f32,np.float64), T32_only=(np.float64,np.float64,f32), rhs32_only=(np.float64,np.float64,None,f32)).items():
        c=pipeline(wp[b],tsb,h,t,*args)
        sc=np.abs(ref).max(axis=(0,2),keepdims=True)
        e=(np.abs(c-ref)/sc).max(axis=(0,2))
        res.setdefault(name,[]).append(e)
for k,v in res.items():
    v=np.array(v)
    print(k,'median per coeff order',np.median(v,0).round(9),'max',v.max(0).round(8))
