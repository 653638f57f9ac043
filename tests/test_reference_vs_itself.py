"""
G6: how far does the REFERENCE part from ITSELF?

tests/golden/g6_reference_vs_itself.npz holds the finals of the real `MinJerkPlanner.plan_once`
(expert_planner.py:205-237, run read-only by tools/gen_golden.py) on 256 M = 21 requests of the 2-D reference map (64 in round 3; requests 0..63 unchanged), under
four BLAS environments that differ by environment variables only -- no source change:

    blas_threads_1             OPENBLAS_NUM_THREADS=1  (the environment of every other fixture)
    blas_threads_8             OPENBLAS_NUM_THREADS=8
    blas_coretype_haswell      OPENBLAS_CORETYPE=Haswell      (another kernel set of the same OpenBLAS: other FMA / blocking)
    blas_coretype_sandybridge  OPENBLAS_CORETYPE=Sandybridge

The objective is evaluated through `np.linalg.solve` (LAPACK dgesv, :336 / :503); a different kernel set changes its
last bits.  Recorded outcome: the thread count changes nothing (these 126 x 126 systems run single-threaded either way);
a different kernel set leaves NO final bit-identical and parts 11 % of the runs beyond north_star's 1e-4 -- the reference
against itself.  That is the yardstick for every whole-run comparison in this repository.

CPU: the fixture's own statistics; the C++ oracle against the reference's finals, judged by that yardstick.
GPU: the device's fp64 parity mode against the reference's finals, judged by the same yardstick.
"""
import contextlib
import io

import numpy as np
import pytest

from helpers import golden, load

NQ = 40          # D (M - 1) = 2 * 20 control-point coordinates
BASE = "blas_threads_1"


def _fixture():
    return load(golden("g6_reference_vs_itself.npz")[0])


def _finals(d, env):
    n = int(d["n_requests"])
    x = np.stack([d[f"{env}__q{k}_x"] for k in range(n)])
    nfev = np.array([int(d[f"{env}__q{k}_nfev"]) for k in range(n)])
    err = [str(d[f"{env}__q{k}_error"]) for k in range(n)]
    return x, nfev, err


def _dx(x, ref):
    return np.abs(x[:, :NQ] - ref[:, :NQ]).max(axis=1) / np.abs(ref[:, :NQ]).max(axis=1)


def _ok(d):
    """requests on which the reference run ended through minimize() (no OverflowError inside a callback)"""
    _, nfev, _ = _finals(d, BASE)
    return nfev > 0


def reference_self_agreement(d):
    """share of the runs on which the reference under another BLAS kernel set ends within 1e-4 of itself"""
    ok = _ok(d)
    xb, nb, _ = _finals(d, BASE)
    out = {}
    for env in ("blas_coretype_haswell", "blas_coretype_sandybridge"):
        x, nf, _ = _finals(d, env)
        sel = ok & (nf > 0)
        out[env] = float((_dx(x[sel], xb[sel]) <= 1e-4).mean())
    return out


def test_thread_count_does_not_matter_but_the_kernel_set_does():
    d = _fixture()
    ok = _ok(d)
    assert ok.sum() >= 60
    xb, nb, eb = _finals(d, BASE)
    x8, n8, e8 = _finals(d, "blas_threads_8")
    assert np.array_equal(x8[ok], xb[ok]) and np.array_equal(n8[ok], nb[ok]) and e8 == eb
    for env in ("blas_coretype_haswell", "blas_coretype_sandybridge"):
        x, nf, er = _finals(d, env)
        sel = ok & (nf > 0)
        dx = _dx(x[sel], xb[sel])
        bit = np.array([np.array_equal(a, b) for a, b in zip(x[sel], xb[sel])])
        assert not bit.any()                                   # every final differs in its last bits ...
        assert np.median(dx) < 1e-12                            # ... most by round-off only ...
        parted = (dx > 1e-4).mean()
        assert 0.05 <= parted <= 0.25, parted                   # ... and about one run in nine beyond north_star's 1e-4
        assert (nf[sel] != nb[sel]).mean() >= 0.05              # with another evaluation count
    assert float(np.mean(nb[ok])) > 60                          # runs of ~100 evaluations, cfg2's length


def _requests(d):
    n = int(d["n_requests"])
    return [(int(d[f"q{k}_map_seed"]), d[f"q{k}_head"], d[f"q{k}_tail"], d[f"q{k}_init_wpts"], d[f"q{k}_init_ts"])
            for k in range(n)]


def test_cpp_oracle_parts_from_the_reference_no_more_than_the_reference_from_itself():
    from oracle import cpu_native as cn
    from oracle import minco_np as onp
    d = _fixture()
    ok = _ok(d)
    xb, nb, _ = _finals(d, BASE)
    maps = {}
    finals = np.full_like(xb, np.nan)
    for k, (ms, head, tail, wp, ts) in enumerate(_requests(d)):
        if not ok[k]:
            continue
        if ms not in maps:
            occ = d[f"occ{ms}"]
            maps[ms] = cn.NativeMap.from_grid2d(onp.GridESDF(occ, float(d["res"]), 300, 300, d["origin"]))
        pl = cn.NativePlanner(onp.PlannerParams())
        pl.read_planning_conditions(maps[ms], head, tail, wp, ts)
        try:
            pl.plan_once()
        except (ValueError, OverflowError):
            pass
        if pl.last_result is not None:
            finals[k] = pl.last_result.x
    sel = ok & np.isfinite(finals[:, 0])
    within = float((_dx(finals[sel], xb[sel]) <= 1e-4).mean())
    yard = min(reference_self_agreement(d).values())
    slack = 2.0 * np.sqrt(0.25 / sel.sum())
    assert within >= yard - slack, (within, yard)


@pytest.mark.gpu
def test_device_parts_from_the_reference_no_more_than_the_reference_from_itself():
    """the fp64 parity mode through the reference-shaped API on the 64 requests: the share of finals within 1e-4 of the
    reference's is at least the reference's own self-agreement under another BLAS kernel set (minus binomial slack)"""
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    d = _fixture()
    ok = _ok(d)
    xb, nb, eb = _finals(d, BASE)
    maps = {}
    finals = np.full_like(xb, np.nan)
    nfev = np.zeros(len(xb), dtype=int)
    same_exc = 0
    for k, (ms, head, tail, wp, ts) in enumerate(_requests(d)):
        if not ok[k]:
            continue
        if ms not in maps:
            maps[ms] = npa.ESDF()
            maps[ms].occupancy_map_cb(synth.OccupancyGridMsg(d[f"occ{ms}"], float(d["res"]), d["origin"]))
        pl = npa.MinJerkPlanner(npa.PlannerConfig())
        pl.read_planning_conditions(maps[ms], head, tail, wp, ts)
        err = ""
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                pl.plan_once()
        except Exception as ex:
            err = f"{type(ex).__name__}:{ex}"
        same_exc += err.split(":")[0] == eb[k].split(":")[0]
        if hasattr(pl, "tau") and not err.startswith("OverflowError"):
            finals[k] = np.concatenate([np.reshape(pl.int_wpts, -1), pl.tau])
            nfev[k] = pl.last_nfev
    sel = ok & np.isfinite(finals[:, 0])
    dx = _dx(finals[sel], xb[sel])
    within = float((dx <= 1e-4).mean())
    yard = reference_self_agreement(d)
    slack = 2.0 * np.sqrt(0.25 / sel.sum())
    print(f"device fp64 vs reference: within 1e-4 on {within:.3f} of {sel.sum()} runs (same nfev {(nfev[sel] == nb[sel]).mean():.3f}, "
          f"median {np.median(dx):.1e}); reference vs itself under another BLAS kernel set: {yard}")
    assert within >= min(yard.values()) - slack, (within, yard)
    assert np.median(dx) < 1e-9
    assert same_exc >= 0.9 * ok.sum()
