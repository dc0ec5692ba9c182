#!/usr/bin/env python
"""Where a kernel's spilled registers are touched: scratch operations per innermost loop of its ISA.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --offload-device-only -S -I include -I juicer_amd/csrc -o /tmp/k.s juicer_amd/csrc/jd_device.hip
   python tools/spillmap.py /tmp/k.s _Z6k_slotILi3EE"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
m = re.search(r'^(%s[^\n]*):' % re.escape(sys.argv[2]), txt, re.M)
i = m.start(); j = txt.index('.Lfunc_end', i)
lines = txt[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = n
loops = []
for n, l in enumerate(lines):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < n:
            loops.append((labels[t], n))
sp = [n for n, l in enumerate(lines) if 'scratch_' in l]
print(m.group(1)[:70], len(lines), 'lines,', len(sp), 'scratch operations,', len(loops), 'loops')


def innermost(n):
    best = None
    for a, b in loops:
        if a <= n <= b and (best is None or (b - a) < (best[1] - best[0])):
            best = (a, b)
    return best


c = Counter(innermost(n) for n in sp)
for k, v in sorted(c.items(), key=lambda kv: (kv[0] is None, kv[0] or (0, 0))):
    print('  loop lines %s (%s long): %d scratch operations' % (k, (k[1] - k[0]) if k else '-', v))
