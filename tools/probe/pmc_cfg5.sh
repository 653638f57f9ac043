cd /tmp && export TMPDIR=/tmp
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES"; do
  d=/root/repo/gpurun_out/pmc5_$(echo $g | cut -c4-14)
  rocprofv3 --pmc $g --output-format csv -d $d -- python3 /root/repo/bench.py --config cfg5 --no-cpu --steps 2 --warmup 1 > /root/repo/gpurun_out/pmc5.log 2>&1
done
python3 - <<PY
import csv,glob,json
agg={}
for fn in glob.glob("/root/repo/gpurun_out/pmc5_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "optimize_kernel" in r["Kernel_Name"]:
            e=agg.setdefault(r["Counter_Name"],[0.0,0]); e[0]+=float(r["Counter_Value"]); e[1]+=1
res={k:v[0]/v[1] for k,v in agg.items()}
line=[l for l in open("/root/repo/gpurun_out/pmc5.log") if l.startswith('{"metric"')]
j=json.loads(line[-1]); ev=j["roofline"]["evals_per_launch"]
print("evals/launch", ev, "samples/eval", j["roofline"]["samples_per_launch"]/ev)
print({k: round(v/ev,1) for k,v in res.items()})
print("VALU active / wave cycles", res["SQ_ACTIVE_INST_VALU"]/res["SQ_WAVE_CYCLES"], "wait", res["SQ_WAIT_ANY"]/res["SQ_WAVE_CYCLES"])
PY
