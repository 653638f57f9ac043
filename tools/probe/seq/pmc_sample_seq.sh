#!/bin/bash
# counters per wavefront of the ESDF-lookup kernel's 4096 launch: the round-4 body (old32), the sequence body (seq32) and --
# with an experiment library in NEO_PLANNER_LIB -- its SEQ_FLAGS variants.   bash tools/probe/pmc_sample_seq.sh "old32 seq32" [outfile]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${2:-gpurun_out/pmc_sample_seq.txt}
for mode in $1; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    d=/tmp/pmc_$RANDOM
    rocprofv3 --pmc $grp --output-format csv -d $d -- python3 tools/experiments/gpu_sample_seq.py --reps 5 --only $mode > /dev/null 2>&1
    python3 - $d "$mode SEQ_FLAGS=${SEQ_FLAGS:-0}" <<'PY' >> $OUT
import csv,glob,sys
agg={}
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sample_" in row["Kernel_Name"] and "kernel" in row["Kernel_Name"]:
            e=agg.setdefault(row["Counter_Name"],[0.0,0]); e[0]+=float(row["Counter_Value"]); e[1]+=1
print(sys.argv[2],{k:round(v[0]/v[1]/4096,1) for k,v in agg.items()})
PY
  done
done
cat $OUT
