#!/usr/bin/env python3
"""SGPR spill traffic of optimize_kernel by NEO_MARK region (listing built with -DNEO_MARKS -S): v_writelane into / v_readlane
out of the vector registers the allocator uses as scalar spill space.  Every reload inside the optimiser loop is a
vector-pipe instruction.   python3 tools/probe/spill_reloads.py /tmp/marks.s 'Li2EfNS_5Map3D.*WaveLanesPD'"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3neo") and "optimize" in l and ":" in l and pat.search(l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
spill_regs = set(re.search(r"v_writelane_b32 (v\d+)", l).group(1) for l in body if "v_writelane_b32" in l)
region = "prologue"
rl, wl = collections.Counter(), collections.Counter()
for l in body:
    m = re.search(r"; NEOMARK (\w+)", l)
    if m:
        region = m.group(1)
    m = re.search(r"v_readlane_b32 s\d+, (v\d+)", l)
    if m and m.group(1) in spill_regs:
        rl[region] += 1
    if "v_writelane_b32" in l:
        wl[region] += 1
print("spill registers", sorted(spill_regs))
print("reloads by region", dict(rl), "total outside the prologue", sum(v for k, v in rl.items() if k != "prologue"))
print("spills by region", dict(wl))
