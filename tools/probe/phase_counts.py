#!/usr/bin/env python3
"""Dynamic instruction counts of the all-fp32 optimiser kernel per evaluation with single phases switched off
(neo_params.flags debug bits: 1 sample loop, 8 joint factorisation, 16 joint solves / scans, 4 L-BFGS history), GPU box:

    python3 tools/probe/phase_counts.py gpurun_out/phase_counts

Needs a -DNEO_EXPERIMENTS build of the library (NEO_PLANNER_LIB): the product's kernels have no phase switches.
The differences to the full kernel price the phases (the runs with a phase off optimise garbage, so line-search
lengths shift a little: read the numbers as estimates)."""
import csv, glob, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1]); os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
for flags in (0, 1):
    env = dict(os.environ, NEO_BENCH_FLAGS_OR=str(flags))
    d = os.path.join(out, f"pmc_{flags}")
    log = os.path.join(out, f"{flags}.log")
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "--output-format", "csv", "-d", d, "--",
           "python3", "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-modes", "--batches-per-step", "4"]
    with open(log, "w") as f:
        rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=300, cwd=REPO, env=env).returncode
    line = [l for l in open(log) if l.startswith('{"metric"')]
    j = json.loads(line[-1]) if line else {}
    evals = j.get("roofline", {}).get("evals_per_launch")
    agg = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "optimize_kernel" in row["Kernel_Name"] and int(row["Grid_Size"]) // 64 == 4096:
                e = agg.setdefault(row["Counter_Name"], [0.0, 0]); e[0] += float(row["Counter_Value"]); e[1] += 1
    res = {k: v[0] / v[1] for k, v in agg.items()}
    print("flags", flags, "rc", rc, "evals/launch", evals, "samples/eval", (j.get("roofline", {}).get("samples_per_launch") or 0) / max(evals or 1, 1),
          {k: round(v / evals, 1) for k, v in res.items()} if evals else res, "traj/s", j.get("value"), flush=True)
