#!/usr/bin/env python3
"""Collects the rocprofv3 evidence for a round on a GPU box (run through gpurun):

  1. `rocprofv3 --kernel-trace --stats` of a short `bench.py` run  -> <out>/kernel_stats.csv
  2. counter passes (`--pmc`, one group per run, nothing else traced) -> <out>/pmc.json
     (mean per dispatch of every counter for the kernels of the hot path)

    python tools/collect_profiles.py gpurun_out/prof_rNN [--config cfg2]

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
]
KERNELS = ["optimize_group_kernel", "optimize_kernel", "sample_kernel", "eval_kernel", "edt3_x", "edt3_y", "edt3_z", "pack3d"]


def run(cmd, log, timeout):
    with open(log, "w") as f:
        try:
            return subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd=REPO).returncode
        except subprocess.TimeoutExpired:
            f.write("\nTIMEOUT\n")
            return 124


def main():
    out = os.path.abspath(sys.argv[1])
    extra = sys.argv[2:]
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    bench = ["python3", "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu"] + extra
    d = os.path.join(out, "trace")
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + bench,
             os.path.join(out, "trace.log"), 420)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
    print("kernel trace rc", rc, "stats", bool(stats))
    agg = {}
    for gi, group in enumerate(PMC_GROUPS):
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
                    e = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                    e[0] += float(row["Counter_Value"])
                    e[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    res = {"command": "rocprofv3 --pmc <group> --output-format csv -- " + " ".join(bench) + "  (one run per group)",
           "groups": PMC_GROUPS,
           "units": {"FETCH_SIZE": "KB per dispatch (raw counter; the 8-byte gathers are an uncalibrated access shape, "
                                   "no gfx950 correction applied)", "WRITE_SIZE": "KB per dispatch",
                     "SQ_*": "summed over the dispatch"},
           "kernels": {k: {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
                       for k, cs in agg.items()}}
    with open(os.path.join(out, "pmc.json"), "w") as f:
        json.dump(res, f, indent=1)
    shutil.rmtree(os.path.join(out, "trace"), ignore_errors=True)


if __name__ == "__main__":
    main()
