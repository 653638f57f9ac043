"""CPU tests of the host side: the C-ABI library builds, loads and exports every symbol the
header declares (no compute calls), and the host-only logic of the reference-shaped classes."""
import os
import re

import numpy as np
import pytest

from conftest import REPO
from helpers import golden, load

import neo_planner_amd as npa
from neo_planner_amd import _lib, build, synth


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(REPO, "include", "neo_planner.h")).read()
    declared = set(re.findall(r"\b(neo_[a-z0-9_]+)\s*\(", header))
    declared -= {"neo_ctx"}
    assert declared, "no declarations found"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.neo_abi_version() == 1


def test_params_default_matches_the_ros_yaml(lib):
    import ctypes
    p = _lib.NeoParams()
    assert lib.neo_params_default(ctypes.byref(p)) == 0
    assert (p.v_max, p.T_min, p.T_max, p.safe_dis, p.delta_t) == (1.0, 0.5, 5.0, 0.7, 0.1)
    assert list(p.weights) == [1.0, 1.0, 1.0, 10000.0]
    assert (p.collision_cost_tol, p.ftol, p.gtol, p.maxls) == (5.0, 1e-4, 1e-4, 20)
    assert p.bugcompat_stale_T == 1


def test_workspace_size(lib):
    # the L-BFGS history lives in LDS: no HBM workspace
    assert lib.neo_optimize_workspace_bytes(4096, 21, 3) == 0


def test_null_context_is_rejected(lib):
    assert lib.neo_params_set(None, None) != 0
    assert lib.neo_ctx_synchronize(None) != 0
    assert lib.neo_last_error(None) == b"null context"


def test_init_variables_match_reference_g4():
    d = load(golden("g4_init.npz")[0])
    for k in range(int(d["n_cases"])):
        pl = npa.MinJerkPlanner(npa.PlannerConfig(init_wpts_mode=str(d[f"c{k}_mode"])))
        w, ts = pl.generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"])
        assert np.array_equal(w, d[f"c{k}_wpts"]) and np.array_equal(ts, d[f"c{k}_ts"])
        np.random.seed(77 + k)
        w2, _ = pl.generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"], seed=2)
        assert np.array_equal(w2, d[f"c{k}_wpts_seeded"])
        if f"c{k}_batch_wpts" in d.files:
            bw, bts = pl.batch_generate_init_variables(d[f"c{k}_head"], d[f"c{k}_tail"])
            assert np.array_equal(bw, d[f"c{k}_batch_wpts"]) and np.array_equal(bts, d[f"c{k}_batch_ts"])


def test_time_map_round_trip_and_errors():
    pl = npa.MinJerkPlanner()
    pl.M = 4
    ts = np.array([0.6, 2.5, 3.75, 4.9])
    assert np.allclose(pl.map_tau2T(pl.map_T2tau(ts)), ts, rtol=1e-15)
    with np.errstate(divide="ignore"):               # ts == T_min -> tau = -inf
        assert pl.map_T2tau(np.array([0.5, 1, 1, 1]))[0] == -np.inf
    with pytest.raises(OverflowError):              # math.exp overflow (:481)
        pl.map_tau2T(np.array([-800.0, 0, 0, 0]))


def test_batch_pack_unpack():
    bp = npa.BatchPlanner()
    head, tail, wp, ts = synth.replan_requests(3, 5, 20, D=3)
    x = bp.pack_x(wp, ts)
    assert x.shape == (5, 3 * 20 + 21)
    w2, t2 = bp.unpack_x(x, 21, 3)
    assert np.array_equal(w2, wp) and np.allclose(t2, ts, rtol=1e-14)


def test_synthetic_forest_is_deterministic_and_clear():
    a = synth.forest_boxes(4)
    assert a == synth.forest_boxes(4) and len(a) in (10, 15, 20)
    for i, (cx, cy, sx, sy, sz) in enumerate(a):
        assert 3 <= cx <= 27 and -5 <= cy <= 5 and 0.5 <= sx <= 1.5 and 3 <= sz <= 6
        for (bx, by, bsx, bsy, _) in a[:i]:
            assert not (abs(cx - bx) < (sx + bsx) / 2 + 1.8 and abs(cy - by) < (sy + bsy) / 2 + 1.8)
    occ = synth.occupancy_2d(4)
    assert occ.shape == (300, 300) and set(np.unique(occ)) <= {0, 100}


def test_flag_constants_match_the_header():
    """NEO_FLAG_* in include/neo_planner.h and in the ctypes binding"""
    import re
    hdr = open(os.path.join(REPO, "include", "neo_planner.h")).read()
    defs = dict(re.findall(r"#define\s+(NEO_FLAG_\w+)\s+(\d+)", hdr))
    assert int(defs["NEO_FLAG_ONE_WAVE_PER_SIMD"]) == _lib.NEO_FLAG_ONE_WAVE_PER_SIMD
    assert int(defs["NEO_FLAG_TWO_WAVES_PER_SIMD"]) == _lib.NEO_FLAG_TWO_WAVES_PER_SIMD
    assert [f for f, _ in _lib.NeoParams._fields_][-1] == "flags"


def test_one_source_for_the_lane_group_kernel():
    """the several-trajectories-per-wavefront kernel instantiates the same device functions as the default kernel
    (csrc/neo_device.hpp templated on the lane-group policy): no generated or hand-kept second copy of the numerics"""
    csrc = os.path.join(REPO, "neo-planner_amd", "csrc")
    assert not os.path.exists(os.path.join(csrc, "neo_group.hpp"))
    dev = open(os.path.join(csrc, "neo_device.hpp")).read()
    grp = open(os.path.join(csrc, "neo_group_kernel.hpp")).read()
    for fn in ("minco_forward", "minco_sample", "minco_backward"):
        assert dev.count(f" {fn}(") == 1, fn                      # defined once ...
        assert f"{fn}<" in grp and "GroupLanes<W>" in grp         # ... and instantiated for the groups
    assert "struct WaveLanes" in dev and "struct GroupLanes" in dev


def test_build_cache_is_keyed_by_content_not_by_time_stamps(tmp_path, monkeypatch):
    """neo_planner_amd/build.py: the library is up to date when its stamp equals the hash of command lines, sources and
    headers -- touching a file changes nothing, changing a header or an option changes every key (no hipcc run here)"""
    import shutil
    from neo_planner_amd import build as b
    if os.path.exists(b.STAMP) and os.path.exists(b.LIB):
        assert open(b.STAMP).read().strip() == b._key()        # the in-tree library is the one these sources build
    k_lib, k_unit = b._key(), b._key("neo_disp_sample.hip")
    hdr = os.path.join(b.CSRC, "neo_linesearch.hpp")
    st = os.stat(hdr)
    os.utime(hdr, (st.st_atime, st.st_mtime + 1000))           # a newer time stamp ...
    try:
        assert b._key() == k_lib and b._key("neo_disp_sample.hip") == k_unit   # ... is no change
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
    copy = tmp_path / "neo_linesearch.hpp"
    shutil.copy(hdr, copy)
    with open(copy, "a") as f:
        f.write("// edited\n")
    real = b._headers
    monkeypatch.setattr(b, "_headers", lambda src=None: [str(copy) if os.path.basename(d) == "neo_linesearch.hpp" else d for d in real(src)])
    assert b._key() != k_lib and b._key("neo_disp_sample.hip") != k_unit       # a changed header: every key changes
    monkeypatch.setattr(b, "_headers", real)
    monkeypatch.setenv("NEO_FP_CONTRACT", "fast")
    assert b._key() != k_lib                                                    # and so does a changed option
