#!/usr/bin/env python3
"""Static instruction counts of optimize_kernel between the NEO_MARK position markers (build with -DNEO_MARKS -S):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include --cuda-device-only -DNEO_MARKS \
          -DNEO_SLIM_BUILD -S neo-planner_amd/csrc/neo_disp_opt3d_x.hip -o /tmp/marks.s
    python3 tools/probe/mark_counts.py /tmp/marks.s 'Li2EfNS_5Map3D.*WaveLanesPD'

Prints, per region (the code after a marker, in listing order), the vector / scalar / LDS instructions by loop depth.
Depth 1 is the optimiser loop (once per evaluation); deeper blocks are the loops inside a phase -- multiply by the trip
counts (cfg2: 5 reduction levels, ~4 sample rounds, 10 history pairs).  Listing order is not execution order: blocks the
compiler moved away from their source position are counted where they landed."""
import re, sys
src, pat = sys.argv[1], re.compile(sys.argv[2])
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3neo") and "optimize" in l and ":" in l and pat.search(l))
region, depth, in_label = "prologue", 0, False
order, acc = [], {}
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if l.startswith(".LBB"):
        # (the label's comment may run over the following lines: "Parent Loop ... Depth=1" then "This Inner Loop Header: Depth=2")
        depth = max([int(x) for x in re.findall(r"Depth=(\d)", l)] or [0])
        in_label = True
        continue
    if in_label and t.startswith(";"):
        depth = max([depth] + [int(x) for x in re.findall(r"Depth=(\d)", l)])
        continue
    in_label = False
    if False:
        continue
    m = re.search(r"; NEOMARK (\w+)", l)
    if m:
        region = m.group(1)
        continue
    op = t.split(" ")[0] if t else ""
    kind = None
    if op.startswith("v_"):
        kind = "valu"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): sub = "lane"
        elif "dpp" in l: sub = "dpp"
        elif op.startswith("v_mov"): sub = "mov"
        elif op.startswith(("v_cndmask", "v_cmp")): sub = "sel"
        elif "_f64" in op: sub = "f64"
        else: sub = "alu"
    elif op.startswith("ds_"): kind, sub = "lds", "lds"
    elif op.startswith(("buffer_", "global_", "scratch_")): kind, sub = "vmem", "vmem"
    elif op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt")): kind, sub = "salu", "salu"
    if kind is None:
        continue
    key = (region, depth)
    if key not in acc:
        acc[key] = {}
        order.append(key)
    acc[key][sub] = acc[key].get(sub, 0) + 1
cols = ["alu", "mov", "dpp", "sel", "lane", "f64", "lds", "vmem", "salu"]
print(f"{'region':18s} depth " + " ".join(f"{c:>5s}" for c in cols) + "   valu")
for key in order:
    a = acc[key]
    valu = sum(a.get(c, 0) for c in ["alu", "mov", "dpp", "sel", "lane", "f64"])
    print(f"{key[0]:18s} {key[1]:5d} " + " ".join(f"{a.get(c, 0):5d}" for c in cols) + f"  {valu:5d}")
