#!/usr/bin/env python3
"""Which noise makes runs end ABOVE the exact run's cost?  cpu_native (the C++ fp64 restatement) on cfg2-shaped requests
of a 3-D scene, against itself with (a) fp32 sampled terms, (b) coefficients perturbed by a relative eps, (c) gradient
entries perturbed by a relative eps, (d) all three at the all-fp32 kernels' levels: run-by-run ratio of final costs
(quantiles, geometric mean, shares above / below by 1e-3) and mean evaluations.  CPU only.

    python tools/experiments/cpu_noise_bias.py [N]"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
import numpy as np
from neo_planner_amd import synth
from oracle import cpu_native as cn
from oracle import minco_np as onp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
grid, M, D = 150, 21, 3
res = 30.0 / grid
dist = synth.esdf_3d(0, n=grid, res=res, canopy=80).astype(np.float32)
head, tail, wp, ts = synth.replan_requests(0, N, M - 1, D=D, **synth.VOLUME)
nm = cn.NativeMap.from_field3d(dist, res, synth.DOMAIN_ORIGIN)
cfg = onp.PlannerParams()
tau = -np.log((cfg.T_max - cfg.T_min) / (ts - cfg.T_min) - 1.0)
x0 = np.concatenate([wp.reshape(N, -1), tau], axis=1)
w = np.asarray(cfg.weights)
thr = os.cpu_count() or 1
base = cn.optimize_batch(nm, x0, head, tail, M, D, threads=thr)
c0 = (base["costs"] * w).sum(axis=1)
ok0 = (base["status"] & 0xff) <= 2
print(f"exact: mean nfev {base['nfev'].mean():.1f}, median cost {np.median(c0):.3f}")
cases = [("fp32 sampled terms", dict(sample_f32=True)),
         ("coefficients 1e-7", dict(coeff_eps=1e-7)), ("coefficients 1e-6", dict(coeff_eps=1e-6)), ("coefficients 5e-6", dict(coeff_eps=5e-6)),
         ("gradient 3e-6", dict(grad_eps=3e-6)), ("gradient 3e-5", dict(grad_eps=3e-5)),
         ("one-ulp coefficients", dict(coeff_eps=2.2e-16)),
         ("all-fp32-like (f32 samples, coeff 1e-7, grad 3e-6)", dict(sample_f32=True, coeff_eps=1e-7, grad_eps=3e-6)),
         ("f32 samples, coeff 5e-6, grad 3e-6", dict(sample_f32=True, coeff_eps=5e-6, grad_eps=3e-6))]
for name, kw in cases:
    o = cn.optimize_batch(nm, x0, head, tail, M, D, params=cn.make_params(**kw), threads=thr)
    c1 = (o["costs"] * w).sum(axis=1)
    sel = ok0 & ((o["status"] & 0xff) <= 2) & np.isfinite(c1) & np.isfinite(c0) & (c0 > 0)
    r = c1[sel] / c0[sel]
    q = np.quantile(r, [0.05, 0.25, 0.5, 0.75, 0.95])
    print(f"{name:52s} n {int(sel.sum()):5d} nfev {o['nfev'].mean():6.1f}  ratio q05..q95 {q[0]:.4f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f} {q[4]:.4f}  "
          f"geo-mean {np.exp(np.mean(np.log(r))):.4f}  above/below 1e-3: {100 * (r > 1.001).mean():.1f} % / {100 * (r < 0.999).mean():.1f} %", flush=True)
