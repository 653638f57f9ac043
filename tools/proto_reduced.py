#!/usr/bin/env python3
"""
Design-validation prototype (NumPy) of the solve the HIP kernels use.

The reference solves a dense 6M x 6M system for the MINCO coefficients and a
transposed one for the adjoint (expert_planner.py:336, :503).  The same c(q,T)
and the same gradients follow from a much smaller system:

  * joint states z_j = (p_j, v_j, a_j); p_j = q_{j-1} is given, y_j = (v_j, a_j)
    are the unknowns at the M-1 interior joints;
  * a quintic piece is fixed by the states at its two ends (Hermite form);
  * continuity of jerk and snap at every interior joint (rows 6i+7, 6i+8 of the
    reference's A) is a block-tridiagonal system with 2x2 blocks in y.

This file checks c, dW/dq, dW/dT (with the reference's stale-T quirk) against
oracle/minco_np.py to round-off.  Run: python tools/proto_reduced.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import minco_np as onp  # noqa: E402


def functionals(T):
    """rows js, ss, je, se: jerk/snap at the start/end of a quintic as linear
    functionals of Z = (p0, v0, a0, p1, v1, a1)."""
    T2, T3, T4 = T * T, T**3, T**4
    js = np.array([-60 / T3, -36 / T2, -9 / T, 60 / T3, -24 / T2, 3 / T])
    ss = np.array([360 / T4, 192 / T3, 36 / T2, -360 / T4, 168 / T3, -24 / T2])
    je = np.array([-60 / T3, -24 / T2, -3 / T, 60 / T3, -36 / T2, 9 / T])
    se = np.array([-360 / T4, -168 / T3, -24 / T2, 360 / T4, -192 / T3, 36 / T2])
    return js, ss, je, se


def hermite_coeffs(T, Z):
    """Z (6, D) -> c (6, D)"""
    p0, v0, a0, p1, v1, a1 = Z
    ep = p1 - p0 - T * v0 - 0.5 * T * T * a0
    ev = v1 - v0 - T * a0
    ea = a1 - a0
    c = np.zeros_like(Z)
    c[0], c[1], c[2] = p0, v0, 0.5 * a0
    c[3] = (10 * ep - 4 * T * ev + 0.5 * T * T * ea) / T**3
    c[4] = (-15 * ep + 7 * T * ev - T * T * ea) / T**4
    c[5] = (6 * ep - 3 * T * ev + 0.5 * T * T * ea) / T**5
    return c


def hermite_T_apply(T, g):
    """gz = H(T)^T g for g (6, D): sensitivity wrt Z given sensitivity wrt c."""
    # c3 = (10 ep - 4T ev + T^2/2 ea)/T^3 etc.; collect d/d(ep,ev,ea)
    gep = 10 * g[3] / T**3 - 15 * g[4] / T**4 + 6 * g[5] / T**5
    gev = -4 * g[3] / T**2 + 7 * g[4] / T**3 - 3 * g[5] / T**4
    gea = 0.5 * g[3] / T - g[4] / T**2 + 0.5 * g[5] / T**3
    gz = np.zeros_like(g)
    gz[0] = g[0] - gep
    gz[1] = g[1] - T * gep - gev
    gz[2] = 0.5 * g[2] - 0.5 * T * T * gep - T * gev - gea
    gz[3] = gep
    gz[4] = gev
    gz[5] = gea
    return gz


def build_joint_system(ts):
    """block tridiagonal K (lists of 2x2) over unknowns y_j = (v_j, a_j), j=1..M-1,
    rows (R1 = jerk jump, R2 = snap jump)."""
    M = len(ts)
    F = [functionals(T) for T in ts]
    Lo, Di, Up = [], [], []
    for j in range(1, M):
        jsb, ssb, _, _ = F[j]          # piece j starts at joint j
        _, _, jea, sea = F[j - 1]      # piece j-1 ends at joint j
        Lo.append(np.array([[jea[1], jea[2]], [sea[1], sea[2]]]))
        Di.append(np.array([[jea[4] - jsb[1], jea[5] - jsb[2]], [sea[4] - ssb[1], sea[5] - ssb[2]]]))
        Up.append(np.array([[-jsb[4], -jsb[5]], [-ssb[4], -ssb[5]]]))
    return F, Lo, Di, Up


def joint_rhs(F, P, head, tail):
    """r_j (2, D): minus the known-state part of the jumps.  P = [p_0 .. p_M] (M+1, D)."""
    M = len(F)
    r = []
    for j in range(1, M):
        jsb, ssb, _, _ = F[j]
        _, _, jea, sea = F[j - 1]
        known1 = jea[0] * P[j - 1] + jea[3] * P[j] - jsb[0] * P[j] - jsb[3] * P[j + 1]
        known2 = sea[0] * P[j - 1] + sea[3] * P[j] - ssb[0] * P[j] - ssb[3] * P[j + 1]
        if j == 1:
            known1 = known1 + jea[1] * head[1] + jea[2] * head[2]
            known2 = known2 + sea[1] * head[1] + sea[2] * head[2]
        if j == M - 1:
            known1 = known1 - jsb[4] * tail[1] - jsb[5] * tail[2]
            known2 = known2 - ssb[4] * tail[1] - ssb[5] * tail[2]
        r.append(-np.stack([known1, known2]))
    return r


def block_thomas(Lo, Di, Up, rhs, transpose=False):
    """solve K y = rhs (or K^T y = rhs) by block Thomas, no pivoting across blocks."""
    n = len(Di)
    if transpose:
        Lo_t = [None] + [Up[j - 1].T for j in range(1, n)]
        Up_t = [Lo[j + 1].T for j in range(n - 1)] + [None]
        Di_t = [d.T for d in Di]
        Lo, Di, Up = Lo_t, Di_t, Up_t
    Dp = [None] * n
    Rp = [None] * n
    Dp[0] = Di[0].copy()
    Rp[0] = rhs[0].copy()
    for j in range(1, n):
        W = Lo[j] @ np.linalg.inv(Dp[j - 1])
        Dp[j] = Di[j] - W @ Up[j - 1]
        Rp[j] = rhs[j] - W @ Rp[j - 1]
    y = [None] * n
    y[n - 1] = np.linalg.solve(Dp[n - 1], Rp[n - 1])
    for j in range(n - 2, -1, -1):
        y[j] = np.linalg.solve(Dp[j], Rp[j] - Up[j] @ y[j + 1])
    return y


def forward(q, ts, head, tail):
    """q (D, M-1) -> coeffs (6M, D), and the intermediate state"""
    M = len(ts)
    D = q.shape[0]
    P = np.vstack([head[0][None], q.T, tail[0][None]])           # (M+1, D)
    F, Lo, Di, Up = build_joint_system(ts)
    y = block_thomas(Lo, Di, Up, joint_rhs(F, P, head, tail)) if M > 1 else []
    Zs = []
    V = [head[1]] + [yy[0] for yy in y] + [tail[1]]
    Acc = [head[2]] + [yy[1] for yy in y] + [tail[2]]
    coeffs = np.zeros((6 * M, D))
    for i in range(M):
        Z = np.stack([P[i], V[i], Acc[i], P[i + 1], V[i + 1], Acc[i + 1]])
        Zs.append(Z)
        coeffs[6 * i:6 * i + 6] = hermite_coeffs(ts[i], Z)
    return coeffs, dict(F=F, Lo=Lo, Di=Di, Up=Up, Zs=Zs, P=P)


def backward(grad_C, grad_T_partial, coeffs, ts, st, stale_T=True):
    """total gradients wrt q (D, M-1) and T (M,) given dW/dc and the direct dW/dT."""
    M = len(ts)
    D = grad_C.shape[1]
    F, Lo, Di, Up, Zs = st["F"], st["Lo"], st["Di"], st["Up"], st["Zs"]
    gz = [hermite_T_apply(ts[i], grad_C[6 * i:6 * i + 6]) for i in range(M)]
    # sensitivity wrt joint states
    S = [None] * (M + 1)
    S[0] = gz[0][0:3]
    for j in range(1, M):
        S[j] = gz[j - 1][3:6] + gz[j][0:3]
    S[M] = gz[M - 1][3:6]
    lam = block_thomas(Lo, Di, Up, [S[j][1:3] for j in range(1, M)], transpose=True) if M > 1 else []
    # lam[j-1] (2, D): multiplier of the (jerk, snap) jump equations at joint j
    grad_q = np.zeros((D, M - 1))
    grad_T = np.array(grad_T_partial, dtype=np.float64).copy()
    G_tail = S[M].copy()
    for j in range(1, M):
        l1, l2 = lam[j - 1]
        jsb, ssb, _, _ = F[j]
        _, _, jea, sea = F[j - 1]
        # residual_j = K y - r ; dW = -lam . d(residual at fixed y)
        # dependence on p_{j-1}, p_j, p_{j+1}
        dp_prev = jea[0] * l1 + sea[0] * l2
        dp_here = (jea[3] - jsb[0]) * l1 + (sea[3] - ssb[0]) * l2
        dp_next = -jsb[3] * l1 - ssb[3] * l2
        if j - 1 >= 1:
            grad_q[:, j - 2] -= dp_prev
        grad_q[:, j - 1] -= dp_here
        if j + 1 <= M - 1:
            grad_q[:, j] -= dp_next
        else:
            G_tail[0] -= dp_next
            G_tail[1] -= -jsb[4] * l1 - ssb[4] * l2
            G_tail[2] -= -jsb[5] * l1 - ssb[5] * l2
    for j in range(1, M):
        grad_q[:, j - 1] += S[j][0]
    # time gradients
    for i in range(M):
        T = ts[i]
        c = coeffs[6 * i:6 * i + 6]
        Z = Zs[i]
        v1, a1 = Z[4], Z[5]
        jerk_end = 6 * c[3] + 24 * T * c[4] + 60 * T * T * c[5]
        snap_end = 24 * c[4] + 120 * T * c[5]
        crackle = 120 * c[5]
        # direct: c = H(T) Z at fixed Z
        grad_T[i] -= np.sum(gz[i][3] * v1 + gz[i][4] * a1 + gz[i][5] * jerk_end)
        js, ss, je, se = F[i]
        if i + 1 <= M - 1:      # joint i+1: this piece ends there  (+je.Z, +se.Z)
            l1, l2 = lam[i]
            d_je = snap_end - (je[3] * v1 + je[4] * a1 + je[5] * jerk_end)
            d_se = crackle - (se[3] * v1 + se[4] * a1 + se[5] * jerk_end)
            grad_T[i] -= np.sum(l1 * d_je + l2 * d_se)
        if i >= 1:              # joint i: this piece starts there  (-js.Z, -ss.Z)
            l1, l2 = lam[i - 1]
            d_js = -(js[3] * v1 + js[4] * a1 + js[5] * jerk_end)
            d_ss = -(ss[3] * v1 + ss[4] * a1 + ss[5] * jerk_end)
            grad_T[i] += np.sum(l1 * d_js + l2 * d_ss)
    if stale_T and M >= 2:
        Tl, Ts = ts[M - 1], ts[M - 2]
        c = coeffs[6 * (M - 1):]

        def dE(T):
            return np.array([[0, 1, 2*T, 3*T**2, 4*T**3, 5*T**4],
                             [0, 0, 2, 6*T, 12*T**2, 20*T**3],
                             [0, 0, 0, 6, 24*T, 60*T**2]])
        grad_T[M - 1] += np.trace(G_tail.T @ (dE(Tl) - dE(Ts)) @ c)
    return grad_q, grad_T, G_tail


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = dict(c=0, gq=0, gT=0, gt=0)
    for trial in range(200):
        M = int(rng.choice([2, 3, 5, 21, 41]))
        D = int(rng.choice([2, 3]))
        ts = rng.uniform(0.5, 5.0, M)
        if trial % 4 == 0:
            ts = rng.choice([0.5001, 4.999], M)
        head = rng.normal(0, 1, (3, D)); tail = rng.normal(0, 1, (3, D)); tail[0] += 10
        q = np.cumsum(rng.normal(1, 0.5, (D, M - 1)), axis=1)
        pl = onp.OraclePlanner(onp.PlannerParams())
        pl.D, pl.M = D, M
        pl.head_state, pl.tail_state = head, tail
        pl.ts = ts
        pl.tau = pl.map_T2tau(ts)
        pl.get_coeffs(q, ts)
        c, st = forward(q, ts, head, tail)
        worst["c"] = max(worst["c"], np.abs(c - pl.coeffs).max() / np.abs(pl.coeffs).max())
        pl.grad_C = rng.normal(0, 1, (6 * M, D))
        pl.grad_T = rng.normal(0, 1, M)
        gq_ref, _ = pl.propagate_grad_q_tau()
        gT_ref = pl.grad_T_total
        gq, gT, Gt = backward(pl.grad_C, pl.grad_T, c, ts, st)
        worst["gq"] = max(worst["gq"], np.abs(gq - gq_ref).max() / np.abs(gq_ref).max())
        worst["gT"] = max(worst["gT"], np.abs(gT - gT_ref).max() / np.abs(gT_ref).max())
        worst["gt"] = max(worst["gt"], np.abs(Gt - pl.G[-3:]).max() / np.abs(pl.G[-3:]).max())
    print("worst relative errors vs oracle:", worst)
