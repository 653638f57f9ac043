#!/bin/bash
# One GPU call: default bench line, the other configurations, then the GPU test suite.  Outputs under gpurun_out/.
tag=${1:-x}
python bench.py > gpurun_out/bench_$tag.json 2>/dev/null
rm -f gpurun_out/bench_${tag}_other.jsonl
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --no-cpu 2>/dev/null | tail -1 >> gpurun_out/bench_${tag}_other.jsonl; done
python bench.py --layout yz4 --no-cpu --no-modes --no-retries 2>/dev/null | tail -1 >> gpurun_out/bench_${tag}_other.jsonl
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
print(d["value"], {k: (round(v["value"]), round(v["accepted_traj_per_s"])) for k, v in d.get("modes", {}).items()})
print("plan_ms", d["cfg1"]["plan_ms_gpu"], d["cfg1"]["batched_replans_per_s_fp64_host_buffers"])
print("esdf", d["esdf_kernel"]["frac_8d2"], d["esdf_kernel"]["whole_step_launch"]["frac_8d2"], "build ms", d["esdf_build"]["ms"])
for l in open("gpurun_out/bench_${tag}_other.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:60], d["config"]["workload"].split("layout ")[1][:6], round(d["value"]), d["ms_per_step"])
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
