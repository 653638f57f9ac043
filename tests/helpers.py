"""shared helpers for the parity tests"""
import glob
import os

import numpy as np

from conftest import GOLDEN


def golden(pattern):
    files = sorted(glob.glob(os.path.join(GOLDEN, pattern)))
    assert files, f"no golden fixtures match {pattern}"
    return files


def load(path):
    return np.load(path, allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / denom)


def reference_jump(d, M, rel_noise=4e-6, trials=12):
    """how far the REFERENCE's own cost moves when x moves by fp32-sized noise: the objective is discontinuous (nearest
    cell lookups, esdf.py:61-62; int(T / delta_t) sample counts, expert_planner.py:401) -- a point within ~1e-4 m of a
    cell face or a duration within 1e-6 of a multiple of delta_t is a point where ANY fp32 evaluation may land on the other
    side (the map gradient is piecewise constant: it jumps at every cell face, also where the distance does not).
    Returns (largest relative change of the cost, of the gradient).  Evaluated with the pinned NumPy oracle (bit-equal to
    the reference on G1)."""
    from oracle import minco_np as onp
    t = f"M{M}_"
    x = d[t + "x"]
    o2 = onp.GridESDF(d["occ"], float(d["res"]), d["occ"].shape[1], d["occ"].shape[0], d["origin"])
    pl = onp.OraclePlanner(onp.PlannerParams())
    pl.read_planning_conditions(o2, d[t + "head"], d[t + "tail"], x[:2 * (M - 1)].reshape(2, M - 1), np.ones(M))
    c0 = pl.get_cost(x)
    g0 = pl.get_grad(x)
    rng = np.random.default_rng(M)
    jump, gjump = 0.0, 0.0
    for _ in range(trials):
        xp = x * (1.0 + rel_noise * rng.standard_normal(x.shape))
        jump = max(jump, abs(pl.get_cost(xp) - c0) / abs(c0))
        gjump = max(gjump, float(np.max(np.abs(pl.get_grad(xp) - g0)) / np.max(np.abs(g0))))
    return float(jump), float(gjump)
