#!/usr/bin/env python
"""Static instruction mix of compiled kernels (dev tool): hipcc -S one .hip source for gfx950 and, for every kernel whose
mangled name contains one of the given substrings, print the counts of MFMA / VALU / SALU / LDS / VMEM instructions, of
branches and of s_waitcnt, and the basic-block structure around the MFMA loop.  A branch or a wait per ELEMENT of an
unrolled epilogue (what the VQ sweep paid per distance until round 6) shows up here before any profiler run.
usage: python tools/isa_mix.py chainer-vq-vae_amd/csrc/vq.hip vq_mfma_x3_kernelILi128ELb0ELi2E [more substrings]"""
import collections, os, re, subprocess, sys

src = sys.argv[1]
pats = sys.argv[2:] or ['']
out = '/tmp/isa_mix_%d.s' % os.getpid()
if src.endswith('.s'):
    out = src
else:
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-S',
                        '--cuda-device-only', src, '-o', out], stderr=subprocess.PIPE, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(1)
text = open(out).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    if not any(p in name for p in pats):
        continue
    c = collections.Counter()
    waits = collections.Counter()
    for x in body:
        x = x.strip()
        if not x or x[0] in ';.' or x.endswith(':'):
            continue
        op = x.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            c['vmem'] += 1
            c['vmem ' + ('store' if 'store' in op else 'load')] += 1
        elif op == 's_waitcnt': c['waitcnt'] += 1; waits[x.split(None, 1)[1] if ' ' in x else ''] += 1
        elif op.startswith('s_cbranch') or op == 's_branch': c['branch'] += 1
        elif op == 's_barrier': c['barrier'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    print(name[:110])
    print('   ', dict(c))
    print('    waits:', dict(waits.most_common(8)))
