#!/bin/bash
# register / scratch footprint of the cfg2 kernels of a slim build: tools/probe/kstats.sh [extra -D flags]
cd "$(dirname "$0")/../../neo-planner_amd/csrc" || exit 1
mkdir -p /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -S --cuda-device-only -w -DNEO_SLIM_BUILD "$@" -o /tmp/isa/slim.s neo_kernels.hip || exit 1
python3 - <<'PY'
import re
txt=open('/tmp/isa/slim.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    name=m.group(1); body=m.group(2)
    if 'optimize_kernel' not in name and 'sample_kernel' not in name: continue
    g=lambda k: re.search(k+r':\s+(\d+)', body)
    short=re.sub(r'.*(optimize_kernel|sample_kernel)ILi3E', r'\1<', name)[:40]
    print(short, 'vgpr', g(r'\.vgpr_count').group(1), 'agpr', g(r'\.agpr_count').group(1) if g(r'\.agpr_count') else '-', 'spill', g(r'\.vgpr_spill_count').group(1), 'scratch', g(r'\.private_segment_fixed_size').group(1))
PY
