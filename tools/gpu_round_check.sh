#!/bin/bash
# One GPU call: default bench line (+ its details file), the other configurations, then the GPU test suite.  Outputs under gpurun_out/.
tag=${1:-x}
python bench.py --steps 20 --warmup 5 --details gpurun_out/bench_${tag}_details.json > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
rm -f gpurun_out/bench_${tag}_other.jsonl
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --no-cpu --details gpurun_out/bench_${tag}_${c}_details.json 2>/dev/null | tail -1 >> gpurun_out/bench_${tag}_other.jsonl; done
python bench.py --layout yz4 --no-cpu --no-modes --no-retries --details gpurun_out/bench_${tag}_yz4_details.json 2>/dev/null | tail -1 >> gpurun_out/bench_${tag}_other.jsonl
python - <<PY
import json
s = open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1]
d = json.loads(s)
print("line bytes", len(s), "value", d["value"], {k: (round(v["value"]), v.get("finals_within_1e_4")) for k, v in d.get("modes", {}).items()})
print("single", d.get("single_batch_traj_per_s"), "roofline", d["roofline"]["frac"], d["roofline"].get("frac_aggregate"), "cpu", d.get("cpu_baseline", {}).get("value"))
print("esdf", d["esdf_kernel"]["frac"], d["esdf_kernel"].get("whole_step_launch", {}).get("frac_8d2"), "build ms", d.get("esdf_build", {}).get("ms"), "cfg1", d.get("cfg1"))
for l in open("gpurun_out/bench_${tag}_other.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:70], d["config"].get("layout"), round(d["value"]), d["ms_per_step"], "esdf", (d.get("esdf_kernel") or {}).get("frac"))
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
