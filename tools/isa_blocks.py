#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s <kernel-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and re.match(r'^_Z\S+:', l)][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
blocks, cur = [], ['entry', []]
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('//'):
        continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append(cur); cur = [t, []]; continue
    if t.startswith('.'):
        continue
    cur[1].append(t)
blocks.append(cur)


def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'


for name, ins in blocks:
    c = collections.Counter(kind(i.split()[0]) for i in ins)
    extra = [' '.join(i.split()[:2]) for i in ins if i.split()[0] == 's_barrier' or i.startswith(('s_cbranch', 's_branch'))]
    print(name, len(ins), dict(c), extra[:5])
