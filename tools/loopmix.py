#!/usr/bin/env python
"""Instruction mix of a kernel's hottest loop (the backward branch spanning the most MFMAs) from a --save-temps .s file
(dev tool): loopmix.py file.s mangled_name_substring..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
for pat in sys.argv[2:]:
    for name in [n for n in names if pat in n]:
        i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
        lines = s[i:j].split('\n')
        labels = {}
        for n, l in enumerate(lines):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m: labels[m.group(1)] = n
        loops = []
        for n, l in enumerate(lines):
            m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                a = labels[m.group(1)]
                loops.append((sum(1 for x in lines[a:n] if 'v_mfma' in x), a, n))
        if not loops: continue
        cnt, a, n = max(loops)
        ins = [l.strip().split()[0] for l in lines[a:n] if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
        c = Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        print(name[-56:], 'loop', len(ins), 'mfma', cnt, 'VALU', valu, 'SALU', sum(v for k, v in c.items() if k.startswith('s_')),
              'ds', sum(v for k, v in c.items() if k.startswith('ds_')), 'vmem', sum(v for k, v in c.items() if k.startswith('buffer_') or k.startswith('global_') or k.startswith('scratch_')))
        print('    ', [(k, v) for k, v in c.most_common(40) if k.startswith('v_') and not k.startswith('v_mfma')][:14])
