#!/bin/bash
# instruction-cache behaviour of the headline kernel (one counter pass)
cd /tmp && export TMPDIR=/tmp
d=/root/repo/gpurun_out/pmc_icache
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $d -- python3 /root/repo/bench.py --no-cpu --no-modes --steps 2 --warmup 1 > /root/repo/gpurun_out/pmc_icache.log 2>&1
python3 - <<PY
import csv,glob
agg={}
for fn in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "optimize_kernel" in r["Kernel_Name"] and int(r["Grid_Size"])//64 == 4096:
            e=agg.setdefault(r["Counter_Name"],[0.0,0]); e[0]+=float(r["Counter_Value"]); e[1]+=1
print({k: v[0]/v[1] for k,v in agg.items()})
PY
tail -2 /root/repo/gpurun_out/pmc_icache.log | cut -c1-200
