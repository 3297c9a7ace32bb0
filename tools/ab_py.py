"""A/B of module-level constants on one box without environment switches: each argument is a ';'-separated list of
assignments `module.NAME=value` ('-' = none) applied before bench.main() runs in a fresh process; two interleaved rounds.
usage: python tools/ab_py.py "<bench args>" "-" "vqvae_amd.wavenet.PB_REDUCE_GROUP=20" ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = r'''
import sys, importlib
sys.path.insert(0, %r); sys.path.insert(0, %r + '/chainer-vq-vae_amd')
for a in %r.split(';'):
    if a.strip() in ('', '-'): continue
    k, v = a.split('=', 1); m, n = k.strip().rsplit('.', 1)
    setattr(importlib.import_module(m), n, eval(v))
import bench
sys.argv = ['bench.py'] + %r.split()
bench.main()
'''
args = sys.argv[1]
for rep in range(2):
    for var in sys.argv[2:]:
        out = subprocess.run([sys.executable, '-c', RUN % (ROOT, ROOT, var, '--no-cpu-baseline --no-kernel-table --steps 20 --warmup 5 ' + args)],
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print('[%s] FAILED: %s' % (var, out.stderr[-400:]))
            continue
        d = json.loads(line[-1])
        print('[%s] %.3f ms/step (with input %s)' % (var, d['ms_per_step'], d.get('ms_per_step_with_input') and round(d['ms_per_step_with_input'], 3)))
        sys.stdout.flush()
