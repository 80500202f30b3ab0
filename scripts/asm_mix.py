#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file: python scripts/asm_mix.py file.s kernel-name-substring"""
import sys
from collections import Counter
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(src) if key in l and l.rstrip().split(';')[0].rstrip().endswith(':') and not l.startswith(('\t', ' '))][0]
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
c, n = Counter(), 0
for l in src[start + 1:end]:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
        continue
    op = t[0]
    n += 1
    c[op.split('_')[0]] += 1
    if 'dpp' in l:
        c['(dpp)'] += 1
    if op.startswith('v_') and ('b64' in op or 'u64' in op or 'mul_lo' in op or 'mad_u64' in op or 'f64' in op):
        c['(half-rate valu)'] += 1
print(src[start][:80])
print(n, 'instructions', dict(c))
