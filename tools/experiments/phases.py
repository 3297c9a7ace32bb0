# dev tool: per-phase workgroup time of the two-tap float32x2 kernels.  Needs the library built with -DVQ_PHASE_TIMING
# (make EXTRA=-DVQ_PHASE_TIMING LIB=...; tools/experiments/visit_phases.sh swaps it in for one run).
import ctypes as C, sys, os
sys.argv = ['bench.py', '--steps', '4', '--warmup', '2', '--no-graph', '--no-cpu-baseline']
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
try:
    bench.main()
except SystemExit:
    pass
sys.path.insert(0, 'chainer-vq-vae_amd')
from vqvae_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 24)()
lib.vqvae_debug_phases(out, 0)
names = {0: 'linear (bwd-data)', 1: 'gate', 2: 'gate-bwd'}
for e in range(3):
    n = out[e * 8 + 4]
    if n:
        p, l, c, ep = (out[e * 8 + i] / n for i in range(4))
        tot = p + l + c + ep
        print('%-18s workgroups %7d: prologue %6.0f  loop %6.0f  cond step %5.0f  epilogue %6.0f ticks (%.1f / %.1f / %.1f / %.1f %%)%s'
              % (names[e], n, p, l, c, ep, 100 * p / tot, 100 * l / tot, 100 * c / tot, 100 * ep / tot,
                 '  [epilogue: loads done after %.0f]' % (out[e * 8 + 5] / n) if out[e * 8 + 5] else ''))
