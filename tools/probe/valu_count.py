#!/usr/bin/env python3
"""Dynamic instruction counts of the optimiser kernel per cost/gradient evaluation (run on a GPU box):

    python3 tools/probe/valu_count.py <out_dir> [lib.so ...]

For every library (default: the in-tree one) one `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES`
pass over a short bench.py run; the counters of the optimize_kernel dispatches are divided by the evaluations the
bench line reports.  With two waves per SIMD the kernel is bound by VALU issue, so this count
is the figure a change to the arithmetic has to move."""
import csv, glob, json, os, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1])
libs = sys.argv[2:] or [""]
os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
for lib in libs:
    tag = os.path.basename(lib).replace(".so", "") or "default"
    env = dict(os.environ)
    if lib:
        env["NEO_PLANNER_LIB"] = os.path.abspath(lib)
    d = os.path.join(out, "pmc_" + tag)
    log = os.path.join(out, tag + ".log")
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "--output-format", "csv",
           "-d", d, "--", "python3", "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu", "--batches-per-step", "4"]
    with open(log, "w") as f:
        try:
            rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=300, cwd=REPO, env=env).returncode
        except subprocess.TimeoutExpired:
            rc = 124
    line = [l for l in open(log) if l.startswith('{"metric"')]
    evals = json.loads(line[-1])["roofline"]["evals_per_launch"] if line else None
    agg = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if "optimize_kernel" in row["Kernel_Name"]:
                e = agg.setdefault(row["Counter_Name"], [0.0, 0])
                e[0] += float(row["Counter_Value"]); e[1] += 1
    res = {k: v[0] / v[1] for k, v in agg.items()}
    print(tag, "rc", rc, "evals/launch", evals, {k: (round(v / evals, 1) if evals else v) for k, v in res.items()}, flush=True)
    subprocess.run(["rm", "-rf", d])
