#!/bin/bash
# A/B of the ESDF-lookup kernel's bodies on one box (tools/gpu_sample_only.py): the in-tree library against the comparison
# build under tools/probe/_build (NEO_BUILD_DEFS=-DNEO_SAMPLE_SHARED_TAILS NEO_BUILD_OUT=.../libneo_shared.so), timing first, then
# counters per wavefront of the 4096 launch.   bash tools/probe/pmc_sample.sh [timing]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LIBS="in-tree $PWD/tools/probe/_build/libneo_shared.so"    # (round 5 also compared a coefficient-table variant: HISTORY.md)
for i in 1 2; do for lib in $LIBS; do
  if [ $lib = in-tree ]; then unset NEO_PLANNER_LIB; else export NEO_PLANNER_LIB=$lib; fi
  python3 tools/gpu_sample_only.py
done; done
for lib in $LIBS; do
  if [ $lib = in-tree ]; then unset NEO_PLANNER_LIB; else export NEO_PLANNER_LIB=$lib; fi
  python3 tools/gpu_sample_only.py --whole --reps 10
done
if [ "$1" = "timing" ]; then exit 0; fi
for lib in $LIBS; do
  if [ $lib = in-tree ]; then unset NEO_PLANNER_LIB; else export NEO_PLANNER_LIB=$lib; fi
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
    d=/tmp/pmc_$RANDOM
    rocprofv3 --pmc $grp --output-format csv -d $d -- python3 tools/gpu_sample_only.py --reps 5 --no-warmup > /dev/null 2>&1
    python3 - $d $(basename $lib) <<'PY'
import csv,glob,sys
agg={}
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sample_kernel" in row["Kernel_Name"]:
            e=agg.setdefault(row["Counter_Name"],[0.0,0]); e[0]+=float(row["Counter_Value"]); e[1]+=1
print(sys.argv[2],{k:round(v[0]/v[1]/4096,1) for k,v in agg.items()})
PY
  done
done
