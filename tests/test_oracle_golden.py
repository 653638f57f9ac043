"""
Pins oracle/minco_np.py to the real reference through the committed fixtures
(tests/golden, written by tools/gen_golden.py from /root/reference).
CPU only.  Tolerances: fp64 round-off of a dense solve with cond <= 2e4.
"""
import numpy as np
import pytest

from helpers import golden, load, rel_err
from oracle import minco_np as onp


def _params(d):
    v_max, T_min, T_max, safe_dis, delta_t = d["params"]
    return onp.PlannerParams(v_max=v_max, T_min=T_min, T_max=T_max, safe_dis=safe_dis, delta_t=delta_t,
                             weights=list(d["weights"]))


@pytest.mark.parametrize("path", golden("g1_eval_s*.npz"))
def test_g1_per_eval(path):
    d = load(path)
    m = onp.GridESDF(d["occ"], float(d["res"]), d["occ"].shape[1], d["occ"].shape[0], d["origin"])
    for M in (3, 21, 41):
        t = f"M{M}_"
        pl = onp.OraclePlanner(_params(d))
        x = d[t + "x"]
        nq = 2 * (M - 1)
        pl.read_planning_conditions(m, d[t + "head"], d[t + "tail"], x[:nq].reshape(2, M - 1), d[t + "ts"])
        cost = pl.get_cost(x)
        assert rel_err(pl.ts, d[t + "ts"]) < 1e-14
        assert rel_err(pl.coeffs, d[t + "coeffs"]) < 1e-10
        assert rel_err(pl.costs, d[t + "costs"]) < 1e-9
        assert abs(cost - d[t + "cost"]) <= 1e-9 * abs(d[t + "cost"])
        grad = pl.get_grad(x)
        assert rel_err(pl.grad_C, d[t + "grad_C"]) < 1e-9
        assert rel_err(pl.grad_T, d[t + "grad_T"]) < 1e-9
        assert rel_err(grad, d[t + "grad"]) < 1e-8


@pytest.mark.parametrize("path", golden("g2_esdf_*.npz"))
def test_g2_esdf(path):
    d = load(path)
    occ = d["occ"]
    m = onp.GridESDF(occ, float(d["res"]), occ.shape[1], occ.shape[0], d["origin"])
    assert np.array_equal(m.esdf_map, d["esdf_map"])
    assert np.array_equal(m.esdf_grad_x, d["esdf_grad_x"])
    assert np.array_equal(m.esdf_grad_y, d["esdf_grad_y"])
    for p, dis, grd, col in zip(d["probe_pts"], d["probe_dis"], d["probe_grad"], d["probe_collision"]):
        assert float(m.get_edt_dis(p)) == dis
        assert [float(v) for v in m.get_edt_grad(p)] == list(grd)
        assert bool(m.has_collision(p)) == bool(col)


def _run_entry(d, pl, m):
    entry = str(d["entry"])
    seed = int(d["np_seed"])
    if seed >= 0:
        np.random.seed(seed)
    err = ""
    try:
        if entry == "plan":
            pl.plan(m, d["head"], d["tail"])
        elif entry == "batch":
            pl.batch_plan(m, d["head"], d["tail"])
        elif entry == "once":
            pl.read_planning_conditions(m, d["head"], d["tail"], d["init_wpts"], d["init_ts"])
            pl.plan_once()
    except Exception as ex:
        err = f"{type(ex).__name__}:{ex}"
    return err


@pytest.mark.parametrize("path", golden("g3_trace_*.npz"))
def test_g3_optimizer_and_g5_eval(path):
    d = load(path)
    occ = d["occ"]
    m = onp.GridESDF(occ, float(d["res"]), occ.shape[1], occ.shape[0], d["origin"])
    pl = onp.OraclePlanner(onp.PlannerParams())
    err = _run_entry(d, pl, m)
    assert err == str(d["error"])
    assert pl.opt_running_times == int(d["opt_running_times"])
    assert pl.iter_num == int(d["iter_num"])
    assert rel_err(pl.int_wpts, d["final_int_wpts"]) < 1e-6
    assert rel_err(pl.ts, d["final_ts"]) < 1e-6
    if "final_cost" in d.files:
        assert abs(pl.final_cost - d["final_cost"]) < 1e-6 * abs(d["final_cost"])
    if "state_cmd_60" in d.files:
        hz = int(d["state_cmd_hz"])
        assert rel_err(pl.get_full_state_cmd(hz), d["state_cmd_60"]) < 1e-6
        assert rel_err(pl.get_pos_array(), d["pos_array"]) < 1e-6
        assert rel_err(pl.get_vel_array(), d["vel_array"]) < 1e-6


def test_g4_init_variables():
    d = load(golden("g4_init.npz")[0])
    for k in range(int(d["n_cases"])):
        pl = onp.OraclePlanner(onp.PlannerParams(init_wpts_mode=str(d[f"c{k}_mode"])))
        w, ts = pl.generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"])
        assert np.array_equal(w, d[f"c{k}_wpts"]) and np.array_equal(ts, d[f"c{k}_ts"])
        np.random.seed(77 + k)
        w2, _ = pl.generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"], seed=2)
        assert np.array_equal(w2, d[f"c{k}_wpts_seeded"])
        if f"c{k}_batch_wpts" in d.files:
            bw, bts = pl.batch_generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"])
            assert np.array_equal(bw, d[f"c{k}_batch_wpts"]) and np.array_equal(bts, d[f"c{k}_batch_ts"])


def test_trilinear_ties_back_to_nearest_2d():
    """on a z-constant grid queried at cell centres the 3-D
    trilinear distance equals the 2-D nearest-cell distance."""
    d = load(golden("g2_esdf_0.npz")[0])
    occ = d["occ"]
    res = float(d["res"])
    m2 = onp.GridESDF(occ, res, occ.shape[1], occ.shape[0], d["origin"])
    vol = np.repeat(m2.esdf_map[None, :, :], 5, axis=0)
    m3 = onp.Grid3DESDF(vol, res, [d["origin"][0], d["origin"][1], 0.0])
    rng = np.random.default_rng(0)
    for _ in range(200):
        r = rng.integers(0, occ.shape[0]); c = rng.integers(0, occ.shape[1])
        p = [d["origin"][0] + (c + 0.5) * res, d["origin"][1] + (r + 0.5) * res, 0.25]
        assert abs(m3.get_edt_dis(p) - m2.get_edt_dis(p[:2])) < 1e-12
        assert abs(m3.get_edt_grad(p)[2]) < 1e-12
