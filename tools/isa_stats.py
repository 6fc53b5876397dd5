#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S listing: python tools/isa_stats.py file.s substring [--dump]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if re.match(r'^_Z\w+:', l) and pat in l]
for name in names:
    i = s.index(name + ':')
    j = s.index('.Lfunc_end', i)
    body = s[i:j]
    ins = [l.strip().split()[0] for l in body.split('\n')
           if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
    c = Counter(ins)
    meta = s[s.index('.amdhsa_kernel ' + name):]
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1)
    sg = re.search(r'\.amdhsa_next_free_sgpr (\d+)', meta).group(1)
    print(name, 'instrs', len(ins), 'vgpr', vg, 'sgpr', sg)
    mem = {k: v for k, v in c.items() if k.startswith(('global', 's_load', 'ds_', 's_barrier', 'buffer', 'scratch', 'flat'))}
    print('  mem:', mem)
    print('  top:', dict(c.most_common(14)))
    if '--dump' in sys.argv:
        print(body)
