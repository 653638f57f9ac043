"""bench.py's final stdout line stays small enough for the driver's parser (round 4's line had grown to
20 KB and BENCH_r04.json came back with `parsed: null`).  The line is a pure function of the full report
(bench.compact_line); here it is built from round 4's own 20 KB report (profiles/r04_j_bench.json) and from a multi-rank
shaped one, and held under bench.LINE_LIMIT bytes with every contract key present."""
import json
import os
import sys

from conftest import REPO

sys.path.insert(0, REPO)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _full_report():
    return json.load(open(os.path.join(REPO, "profiles", "r04_j_bench.json")))


def test_compact_line_from_a_20_kb_report_fits_and_round_trips():
    import bench
    full = _full_report()
    assert len(json.dumps(full)) > 15000                 # the report that broke the parser
    s = bench.compact_line(full, "gpurun_out/bench_details.json")
    assert "\n" not in s and len(s.encode()) <= bench.LINE_LIMIT < 6000, len(s)
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == float(f"{full['value']:.6g}") and line["unit"] == "traj/s" and line["higher_is_better"] is True
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and "workload" in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "frac_aggregate"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference")
    # the figures VERDICT r4 item 1 / 2 ask for beside the contract
    assert line["value_parity_mode"] == float(f"{full['modes']['f64']['value']:.6g}")
    assert set(line["modes"]) == {"f64", "f32", "f32x"}
    for m in line["modes"].values():
        assert "value" in m and "finals_within_1e_4" in m
    assert "frac" in line["esdf_kernel"] and "kernel_us" in line["esdf_kernel"]
    assert "single_batch_traj_per_s" in line and line["details"] == "gpurun_out/bench_details.json"


def test_compact_line_of_a_many_rank_report_fits():
    import bench
    full = _full_report()
    full.update(n_gpus=8, rccl_ranks=8, gather_ok=True, dist_backend="nccl",
                per_rank_traj_per_s=[1234567.890123 + i for i in range(8)])
    full["config"]["workload"] = full["config"]["workload"] * 3       # a wordy workload string does not break the bound either
    s = bench.compact_line(full, None)
    assert len(s.encode()) <= bench.LINE_LIMIT
    line = json.loads(s)
    assert line["n_gpus"] == 8 and line["gather_ok"] is True


def test_compact_line_of_a_bare_report():
    """--no-report / --no-cpu runs: no esdf_kernel, no cpu_baseline, no modes table"""
    import bench
    full = {k: v for k, v in _full_report().items() if k not in ("esdf_kernel", "cpu_baseline", "cpu_native", "parity", "cfg1",
                                                                  "accepted_after_retries", "esdf_build")}
    full["modes"] = {"f32x": full["modes"]["f32x"]}
    line = json.loads(bench.compact_line(full, None))
    for k in CONTRACT:
        assert k in line
    assert "cpu_baseline" not in line and "value_parity_mode" not in line


def test_traffic_figures_come_only_from_a_profile_of_this_workload_and_kernel():
    """`roofline.traffic` is a counter figure from a COMMITTED rocprofv3 profile; it is reported only when
    that profile's command names this run's workload and its recorded kernel symbol is the instantiation this run launches
    (bench.Rank.kernel_symbol spells it as rocprofv3 does), and null otherwise"""
    import types
    import bench
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_report
    prof = json.load(open(os.path.join(REPO, "profiles", "r06_pmc.json")))     # (the profile of THIS tree's kernels)

    def rank(argv):
        a = bench.parse(argv)
        a.argv = list(argv)
        R = types.SimpleNamespace(a=a, M=a.waypoints + 1, D=3, n=3 * a.waypoints + a.waypoints + 1, store="f32")
        R.kernel_symbol = lambda name: bench.Rank.kernel_symbol(R, name)
        return R
    R = rank([])
    for key in ("optimize_kernel@4096", "sample_kernel@4096", "sample_kernel@163840"):
        sym = R.kernel_symbol(key.split("@")[0])
        assert prof["kernel_symbols"][key].replace(" ", "").startswith(sym.replace(" ", "")), (key, sym)
        t = bench_report.kernel_traffic(R, key)
        assert t["traffic"] and t["source"] == "profiles/r06_pmc.json"      # (round 5's profiles name another instantiation of the
                                                                             #  ESDF-lookup kernel: fp64 operand buffers)
    # another arithmetic mode, another layout, another size: no figure
    for argv in (["--dtype", "f64"], ["--layout", "yz4"], ["--batch", "2048"], ["--config", "cfg4"]):
        assert bench_report.kernel_traffic(rank(argv), "optimize_kernel@4096")["traffic"] is None, argv
    # cfg5 has its own profile (three FLAT slots, lane = piece, fp16 bricks)
    R5 = rank(["--config", "cfg5"])
    R5.M, R5.n, R5.store = 41, 161, "f16"
    t5 = bench_report.kernel_traffic(R5, "optimize_kernel@4096")
    assert t5["traffic"] and "cfg5" in t5["source"], t5
    # the same command line but another instantiation (a profile with symbols must agree with the run's)
    R2 = rank([])
    R2.kernel_symbol = lambda name: "void neo::optimize_kernel<3, 2, double"
    t2 = bench_report.kernel_traffic(R2, "optimize_kernel@4096")
    assert t2["source"] is None or ("r05" not in t2["source"] and "r06" not in t2["source"])
