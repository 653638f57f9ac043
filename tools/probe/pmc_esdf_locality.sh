#!/bin/bash
# L2 / fabric counters of sample_kernel for field layouts x dispatch orders (tools/gpu_esdf_locality.py), one rocprofv3
# --pmc pass per counter group and variant; summary -> gpurun_out/pmc_esdf_locality_<cfg>.json
#   tools/probe/pmc_esdf_locality.sh [cfg2|cfg5]
cfg=${1:-cfg2}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_loc_$cfg
rm -rf $out; mkdir -p $out
for lay in yz4 brick; do
  for ord in "index" "spatial 1 m" "sorted, no XCD deal"; do
    tag=$(echo "${lay}_${ord}" | tr ' ,' '__')
    gi=0
    for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum"; do
      (cd /tmp && rocprofv3 --pmc $g --output-format csv -d $out/$tag/g$gi -- python3 $root/tools/gpu_esdf_locality.py --config $cfg --layouts $lay --only-order "$ord" --reps 6 > $out/$tag.g$gi.log 2>&1)
      gi=$((gi+1))
    done
  done
done
python3 - $out $cfg <<'PY'
import csv, glob, json, os, sys
out, cfg = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    agg = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "sample_kernel" not in r["Kernel_Name"]:
                continue
            k = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            e = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            e[0] += float(r["Counter_Value"]); e[1] += 1
    res[os.path.basename(d)] = {str(k): {c: v[0] / v[1] for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(os.path.join(os.path.dirname(out), f"pmc_esdf_locality_{cfg}.json"), "w"), indent=1)
for tag, ks in res.items():
    for k, c in ks.items():
        if "FETCH_SIZE" in c:
            hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
            print(f"{tag:32s} wg {k:>7s}  traffic {(2 * c['FETCH_SIZE'] + c.get('WRITE_SIZE', 0)) * 1024 / 1e6:9.1f} MB  lines {c.get('TCC_EA0_RDREQ_128B_sum', 0):12.0f}  L2 hit {hit:.3f}")
PY
