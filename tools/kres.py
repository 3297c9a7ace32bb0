#!/usr/bin/env python
"""Per-kernel register / LDS / scratch report of a HIP source (dev tool): compiles it with
-Rpass-analysis=kernel-resource-usage and prints the kernels whose name contains one of the given substrings."""
import re, subprocess, sys, os
src = sys.argv[1]
pats = sys.argv[2:] or ['']
extra = os.environ.get('EXTRA', '').split()
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                    '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/tmp/kres_%d.o' % os.getpid()] + extra, capture_output=True, text=True)
txt = r.stderr
if r.returncode:
    print(txt[-3000:]); sys.exit(1)
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if not any(p in dn for p in pats):
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = re.sub(r'\(vq::\w+\)|void vq::', '', dn)
    print('%-56s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s LDS %6s' % (dn[:56], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
