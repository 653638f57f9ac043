#!/bin/bash
# counters of the 3-D EDT kernels (two passes): what the line passes wait for
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; fi
  if [ $pass = 2 ]; then C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU"; fi
  rocprofv3 --pmc $C --output-format csv -d $out/pmc_edt_$pass -- python3 /root/repo/tools/gpu_esdf_build_time.py 300 > $out/pmc_edt_$pass.log 2>&1
done
python3 - <<PY
import csv, glob, json
res = {}
for p in (1, 2):
    for fn in glob.glob(f"$out/pmc_edt_{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            nm = r["Kernel_Name"]
            if "edt3" not in nm and "pack3d" not in nm:
                continue
            key = nm.split("(")[0].replace("void neo::", "").replace("neo::", "")
            e = res.setdefault(key, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
o = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
json.dump(o, open("$out/pmc_edt.json", "w"), indent=1)
for k, d in o.items():
    print(k)
    wc = d.get("SQ_WAVE_CYCLES", 1)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} {v:14.0f}  {v / wc:8.3f} of wave cycles")
PY
