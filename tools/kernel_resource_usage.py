#!/usr/bin/env python3
"""Registers, spills, scratch, occupancy and static LDS of the hot-path kernels, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, every unit with its options from neo_planner_amd/build.py; no GPU needed):

    python tools/kernel_resource_usage.py > profiles/rNN_kernel_resource_usage.txt

Listed: the brick-layout instantiations of the bench's configurations, the budgeted kernels, the ESDF-lookup kernels."""
import concurrent.futures, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neo-planner_amd"))
from neo_planner_amd import build as b

UNITS = ["neo_disp_opt3d_x.hip", "neo_disp_opt3d_w2.hip", "neo_disp_opt3d_f64.hip", "neo_disp_opt3d_b.hip", "neo_disp_sample.hip",
         "neo_disp_group.hip"]


def one(src):
    cmd = [b._hipcc()] + b._flags() + b._unit_flags(src) + ["-c", os.path.join(b.CSRC, src), "-o", os.devnull,
                                                            "-Rpass-analysis=kernel-resource-usage"]
    return src, subprocess.run(cmd, capture_output=True, text=True).stderr


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


print("kernel-resource-usage of the tree at this commit (hipcc -Rpass-analysis=kernel-resource-usage with each unit's options from "
      "neo_planner_amd/build.py);\nbrick-layout instantiations of the bench's configurations, the budgeted kernels, the ESDF-lookup kernels\n")
with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
    results = list(ex.map(one, UNITS))
for src, err in results:
    recs = re.findall(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                      r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", err, re.S)
    names = demangle([r[0] for r in recs])
    print(f"---- {src}  ({' '.join(b._unit_flags(src)) or 'no unit options'})")
    for r, n in zip(recs, names):
        n = re.sub(r"\(.*$", "", n).replace("neo::", "")
        if not (", 3>" in n or "sample_kernel" in n or "group" in n):
            continue
        if "sample_kernel" in n and "Lookup3D" in n and not re.search(r"Lookup3D<\w+, \w+, 3>", n):
            continue
        print(n)
        print(f"    VGPRs {r[2]} AGPRs {r[3]} SGPRs {r[1]} SGPR spills {r[6]} VGPR spills {r[7]} scratch {r[4]} B/lane "
              f"occupancy {r[5]} waves/SIMD LDS {r[8]} B static")
