"""Instruction mix per kernel from the -save-temps ISA (make -C yolo_amd/csrc asm)."""
import re, sys, glob
from collections import Counter
path = glob.glob("build/asm/*gfx950.s")[0]
s = open(path).read()
# kernels start at a label "<mangled>:" and end at s_endpgm
for m in re.finditer(r'^(_ZN2yh\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name = m.group(1)
    body = m.group(2)
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    lines = [l.strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(('.', ';', '/')) and not l.endswith(':')]
    c = Counter(l.split()[0] for l in lines)
    groups = Counter()
    for k, v in c.items():
        g = ('trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', k) else 'valu' if k.startswith('v_') else
             'salu' if k.startswith('s_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
             'lds' if k.startswith('ds_') else 'other')
        groups[g] += v
    print(name[:70], 'total', len(lines), dict(groups))
    if len(sys.argv) > 2:
        print(c.most_common(40))
