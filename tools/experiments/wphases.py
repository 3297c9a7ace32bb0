# dev tool: per-phase step time of the pre-split weight-gradient loops (wgrad3_kernel<4, 1, 2, true, true> register-staged,
# wgrad3_dma_kernel).  Needs a library built with -DVQ_PHASE_TIMING (tools/experiments/abl/lib_<name>.so; usage:
# python tools/experiments/wphases.py <name> ...)
import ctypes as C, sys, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != '--child':
    keep = '/tmp/lib_keep.so'
    shutil.copy(os.path.join(root, 'chainer-vq-vae_amd/libvqvae_hip.so'), keep)
    for n in sys.argv[1:]:
        shutil.copy(os.path.join(root, 'tools/experiments/abl/lib_%s.so' % n), os.path.join(root, 'chainer-vq-vae_amd/libvqvae_hip.so'))
        print('==', n, flush=True)
        subprocess.run([sys.executable, __file__, '--child'], cwd=root)
    shutil.copy(keep, os.path.join(root, 'chainer-vq-vae_amd/libvqvae_hip.so'))
    sys.exit(0)
sys.argv = ['bench.py', '--steps', '4', '--warmup', '2', '--no-graph', '--no-cpu-baseline']
sys.path.insert(0, root)
import bench
try:
    bench.main()
except SystemExit:
    pass
sys.path.insert(0, os.path.join(root, 'chainer-vq-vae_amd'))
from vqvae_amd import _lib
lib = _lib.load()
out = (C.c_ulonglong * 80)()
lib.vqvae_debug_wphases(out, 0)
for k, (name, ph) in enumerate([('register-staged (per pair of steps, first step stamped)', ['fetch issue', 'mma', 'stage (wait + perm + ds_write)', 'barrier']),
                                ('LDS-DMA ring (per step)', ['vmcnt wait', 'barrier', 'issue', 'mma'])]):
    n = out[k * 8 + 4]
    if n:
        v = [out[k * 8 + i] / n for i in range(4)]
        print('%-56s steps %9d: ' % (name, n) + '  '.join('%s %6.0f' % (p, x) for p, x in zip(ph, v)) + '  | sum %6.0f ticks' % sum(v))
n = out[8 + 4]
if n:
    print('per wave of the DMA kernel (ticks per step): wave: vmcnt wait / barrier / issue / mma')
    for w in range(16):
        print('  wave %2d: ' % w + ' / '.join('%6.0f' % (out[16 + 4 * w + i] / n) for i in range(4)))
