#!/bin/bash
# register / scratch footprint of the kernels of one translation unit:
#   tools/probe/kstats.sh [unit.hip ...] [-Dflags]       default unit: neo_disp_opt3d_w2.hip (the two-waves optimiser)
# -DNEO_SLIM_BUILD keeps only the cfg2 instantiation (linear fp32 field) of the optimiser units.
cd "$(dirname "$0")/../../neo-planner_amd/csrc" || exit 1
mkdir -p /tmp/isa
units=(); flags=()
for a in "$@"; do case "$a" in *.hip) units+=("$a");; *) flags+=("$a");; esac; done
[ ${#units[@]} -eq 0 ] && units=(neo_disp_opt3d_w2.hip)
for u in "${units[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../include -S --cuda-device-only -w "${flags[@]}" -o /tmp/isa/${u%.hip}.s $u || exit 1
  python3 - /tmp/isa/${u%.hip}.s <<'PY'
import re, sys
txt=open(sys.argv[1]).read()
meta=txt[txt.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - ', meta)[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', blk) or [None,'-'])[1]
    short=re.sub(r'^_ZN3neo\d+', '', g('name'))[:64]
    print(f"{short:64s} vgpr {g('vgpr_count'):>3} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>3} "
          f"vspill {g('vgpr_spill_count'):>3} sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>5}")
PY
done
