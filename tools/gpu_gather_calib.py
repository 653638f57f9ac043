#!/usr/bin/env python3
"""Calibrates the memory-side counters for the ESDF kernel's access shape and measures what that shape can reach
(tools/probe/gather_calib.hip: 32-byte records, two adjacent 16-byte loads per lane, random or strided), then reads the
same counters for sample_kernel / optimize_kernel from a short bench.py run (GPU box):

    python tools/gpu_gather_calib.py gpurun_out/calib

Counters: FETCH_SIZE (KB), and the L2 -> fabric read requests by size (TCC_EA0_RDREQ_{32B,64B,128B}_sum), whose weighted
sum is the byte count FETCH_SIZE is derived from."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(REPO, "tools", "probe", "_build", "gather_calib")
PASSES = [["FETCH_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]]


def counters(cmd, out, tag, match):
    res = {}
    for gi, group in enumerate(PASSES):
        d = os.path.join(out, f"{tag}_p{gi}")
        with open(os.path.join(out, f"{tag}_p{gi}.log"), "w") as f:
            subprocess.run(["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--"] + cmd, stdout=f,
                           stderr=subprocess.STDOUT, timeout=400, cwd=REPO)
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = match(row)
                if k is None:
                    continue
                e = res.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                e[0] += float(row["Counter_Value"]); e[1] += 1
        shutil.rmtree(d, ignore_errors=True)
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in res.items()}


def true_bytes(c):
    return 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    cases = [("random_432MB_4096w", ["432", "4096", "13", "1", "0"]), ("random_432MB_65536w", ["432", "65536", "13", "1", "0"]),
             ("random_432MB_65536w_independent", ["432", "65536", "13", "0", "0"]),
             ("random_64MB_65536w", ["64", "65536", "13", "1", "0"]), ("random_3MB_65536w", ["3", "65536", "13", "1", "0"]),
             ("stream_stride1_432MB_65536w", ["432", "65536", "13", "0", "1"]),
             ("lines_stride4_432MB_65536w", ["432", "65536", "13", "0", "4"])]
    result = {}
    for name, args in cases:
        p = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        r = json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
        c = counters([BIN] + args, out, name, lambda row: "gather" if "gather" in row["Kernel_Name"] else None).get("gather", {})
        r["counters_per_launch"] = c
        if c and "useful_bytes" in r:
            r["fetch_size_bytes"] = c.get("FETCH_SIZE", 0) * 1024
            r["rdreq_bytes"] = true_bytes(c)
            r["fetch_size_over_useful"] = r["fetch_size_bytes"] / r["useful_bytes"]
            r["rdreq_bytes_over_useful"] = r["rdreq_bytes"] / r["useful_bytes"]
            r["rdreq_GBps"] = r["rdreq_bytes"] / r["kernel_us"] / 1e3
        result[name] = r
        print(name, json.dumps({k: v for k, v in r.items() if k != "counters_per_launch"}), flush=True)
    # the product's kernels under the same counters

    def match(row):
        for k in ("sample_kernel", "optimize_kernel"):
            if k in row["Kernel_Name"]:
                return f"{k}@{int(row['Grid_Size']) // max(int(row['Workgroup_Size']), 1)}"
        return None
    c = counters(["python3", "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-modes"], out, "bench", match)
    for k, v in c.items():
        v["rdreq_bytes"] = true_bytes(v)
        v["fetch_size_bytes"] = v.get("FETCH_SIZE", 0) * 1024
    result["product_kernels"] = c
    print(json.dumps(c), flush=True)
    json.dump(result, open(os.path.join(out, "gather_calib.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
