#!/usr/bin/env python
"""Instruction mix of the loops of one kernel's ISA (no GPU): tools/loopstats.py /tmp/k.s _Z6k_slotILi3EE [min_lines]
(how the ISA is made: see tools/spillmap.py).  Round 6: the slot kernel's passes are bound by what ONE wave can issue, so the
instruction count of a pass loop is a number worth watching when phase A or X is touched."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
m = re.search(r'^(%s[^\n]*):' % re.escape(sys.argv[2]), txt, re.M)
i = m.start(); j = txt.index('.Lfunc_end', i)
lines = txt[i:j].split('\n')
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 300
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = n
loops = set()
for n, l in enumerate(lines):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < n:
            loops.add((labels[t], n))
for a, b in sorted(loops):
    if b - a < minl or b - a > 4000:
        continue
    c = Counter()
    for l in lines[a:b + 1]:
        l = l.strip()
        if not l or l[0] in '.;/' or l.endswith(':'):
            continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith('v_') and k not in ('v_readlane_b32', 'v_writelane_b32'))
    print("loop %5d-%5d: %4d instr | VALU %4d (cndmask %3d, mov %3d) | lane r/w %3d | scalar %4d | saveexec %3d | LDS %3d (b128 %d, bpermute %d) | vmem %3d | waitcnt %3d | nop %3d"
          % (a, b, tot, valu, c['v_cndmask_b32_e64'] + c['v_cndmask_b32_e32'], c['v_mov_b32_e32'] + c['v_mov_b64_e32'],
             c['v_readlane_b32'] + c['v_writelane_b32'],
             sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_and_saveexec', 's_or_saveexec'))),
             c['s_and_saveexec_b64'] + c['s_or_saveexec_b64'], sum(v for k, v in c.items() if k.startswith('ds_')),
             c['ds_read_b128'] + c['ds_read2_b64'], c['ds_bpermute_b32'],
             sum(v for k, v in c.items() if k.startswith(('buffer_', 'global_', 'scratch_'))), c['s_waitcnt'], c['s_nop']))
