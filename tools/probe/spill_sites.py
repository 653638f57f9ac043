#!/usr/bin/env python3
"""where does a kernel spill?  scratch loads / stores and AGPR moves per source line.
usage: spill_sites.py <asm with -gline-tables-only> <regex of the kernel symbol>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2])
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and pat.search(l.split(':')[0]))
end = start
while not lines[end].startswith('.Lfunc_end'):
    end += 1
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
cur = None
cnt = {k: Counter() for k in ('scratch_load', 'scratch_store', 'v_accvgpr_read', 'v_accvgpr_write')}
n_ins = 0
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
    if l.startswith('\t') and not l.strip().startswith(('.', ';')):
        n_ins += 1
    for k in cnt:
        if k in l:
            cnt[k][cur] += 1
print(lines[start][:120])
print('instructions', n_ins, {k: sum(v.values()) for k, v in cnt.items()})
for k, v in cnt.items():
    if v:
        print(k, v.most_common(14))
