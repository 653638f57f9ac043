#!/usr/bin/env python3
"""Collects the rocprofv3 evidence for a round on a GPU box (run through gpurun):

  1. `rocprofv3 --kernel-trace --stats` of a short `bench.py` run  -> <out>/kernel_stats.csv
  2. counter passes (`--pmc`, one group per run, nothing else traced) -> <out>/pmc.json
     (mean per dispatch of every counter for the kernels of the hot path)

    python tools/collect_profiles.py gpurun_out/prof_rNN [--trace-only] [--config cfg2]

rocprofv3 is started as a child process with the program itself after `--`; every run has its own timeout.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_GROUPS = [
    ["FETCH_SIZE"],          # the two TCC-derived sizes do not fit one pass together (rocprofv3 aborts)
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES"],
    ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum"],
    # which pipe is busy: cycles with an instruction of the type executing, per SIMD (quad-cycles,
    # MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units"), next to the CUs' busy cycles
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
     "SQ_BUSY_CU_CYCLES", "SQ_CYCLES"],
    ["SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT",
     "SQ_LDS_IDX_ACTIVE"],
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
     "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
     "SQ_INSTS_VALU_INT64", "SQ_INSTS_SMEM"],
    ["GRBM_GUI_ACTIVE"],
    # L2 -> fabric read requests by size: 32 n32 + 64 n64 + 128 n128 = the bytes FETCH_SIZE is derived from (it reports
    # half of them on gfx950: tools/gpu_gather_calib.py)
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
]
KERNELS = ["optimize_group_kernel", "optimize_kernel", "sample_kernel", "eval_kernel", "edt3_x", "edt3_line", "pack3d"]


def run(cmd, log, timeout):
    with open(log, "w") as f:
        try:
            return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd=REPO).returncode
        except subprocess.TimeoutExpired:
            f.write("\nTIMEOUT\n")
            return 124


def main():
    out = os.path.abspath(sys.argv[1])
    args = sys.argv[2:]
    trace_only = "--trace-only" in args
    # --pmc-groups i,j,...: only these counter groups (indices into PMC_GROUPS; cfg5: instruction mix and pipe activity)
    only = None
    if "--pmc-groups" in args:
        k = args.index("--pmc-groups")
        only = [int(v) for v in args[k + 1].split(",")]
        args = args[:k] + args[k + 2:]
    extra = [a for a in args if a != "--trace-only"]
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    bench = ["python3", "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-modes", "--no-retries",
             "--report-sections", "esdf"] + extra
    d = os.path.join(out, "trace")
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench,
             os.path.join(out, "trace.log"), 420)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
    # the same trace split by launch shape (rocprofv3's own stats merge every launch of a kernel name)
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    symbols = {}          # launch shape -> the kernel's full (demangled) name as rocprofv3 reports it
    if traces:
        by = {}
        with open(traces[0]) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                k = next((k for k in KERNELS if k in name), None)
                if k is None:
                    continue
                wg = max(int(row["Workgroup_Size_X"]) if "Workgroup_Size_X" in row else int(row.get("Workgroup_Size", 64)), 1)
                gs = int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row.get("Grid_Size", 0))
                by.setdefault(f"{k}@{gs // wg}", []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                symbols.setdefault(f"{k}@{gs // wg}", name)
        with open(os.path.join(out, "kernel_stats_by_grid.csv"), "w") as f:
            f.write("kernel@workgroups,calls,avg_ns,min_ns,max_ns,total_ns\n")
            for k, v in sorted(by.items()):
                f.write(f"{k},{len(v)},{sum(v) / len(v):.1f},{min(v):.0f},{max(v):.0f},{sum(v):.0f}\n")
    print("kernel trace rc", rc, "stats", bool(stats), "trace", bool(traces))
    agg = {}
    for gi, group in enumerate([] if trace_only else PMC_GROUPS):
        if only is not None and gi not in only:
            continue
        d = os.path.join(out, f"pmc{gi}")
        rc = run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + bench,
                 os.path.join(out, f"pmc{gi}.log"), 180)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        print("pmc group", group, "rc", rc, "files", len(files))
        for fn in files:
            with open(fn) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    k = next((k for k in KERNELS if k in name), None)
                    if k is None:
                        continue
                    # one entry per launch SHAPE: the stand-alone ESDF kernel runs on 4096 and on 65536 trajectories
                    # in one bench run, and a mean over both belongs to neither
                    wg = max(int(row["Workgroup_Size"]), 1)
                    k = f"{k}@{int(row['Grid_Size']) // wg}"
                    if "duration_ns" not in agg.setdefault(k, {}):
                        agg[k]["duration_ns"] = [0.0, 0]
                    if row["Counter_Name"] == group[0]:
                        agg[k]["duration_ns"][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        agg[k]["duration_ns"][1] += 1
                    e = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                    e[0] += float(row["Counter_Value"])
                    e[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    res = {"command": "rocprofv3 --pmc <group> --output-format csv -- " + " ".join(bench) + "  (one run per group)",
           "groups": PMC_GROUPS,
           "keys": "kernel@workgroups of the launch; duration_ns = mean dispatch duration under the counter passes",
           "units": {"FETCH_SIZE": "KB per dispatch (raw counter; the 8-byte gathers are an uncalibrated access shape, "
                                   "no gfx950 correction applied)", "WRITE_SIZE": "KB per dispatch",
                     "SQ_*": "summed over the dispatch"},
           "kernel_symbols": symbols,
           "kernels": {k: {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
                       for k, cs in agg.items()}}
    with open(os.path.join(out, "pmc.json"), "w") as f:
        json.dump(res, f, indent=1)
    shutil.rmtree(os.path.join(out, "trace"), ignore_errors=True)


if __name__ == "__main__":
    main()
