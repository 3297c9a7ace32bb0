"""Per-launch time of the kernels of one ResidualBlock (gate, res/skip projection, gate-derivative, backward-data,
weight gradients) against the batch size -- i.e. against the number of workgroups in flight (tools/occ_scaling.py does
the same for the dilated conv alone).  usage: python tools/gbwd_scaling.py [dilation]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'chainer-vq-vae_amd')]
from vqvae_amd import _lib, backend as gpu
from vqvae_amd.core import Variable
from vqvae_amd.wavenet import ResidualBlockFunction

gpu.init(0)
lib = _lib.load()
T = 7680
dil = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TAGS = [('gate', _lib.PROF_RESBLOCK_GATE), ('out', _lib.PROF_RESBLOCK_OUT), ('gate-bwd', _lib.PROF_RESBLOCK_BWD_GZ),
        ('bwd-data', _lib.PROF_RESBLOCK_BWD_GX), ('wgrad', _lib.PROF_RESBLOCK_WGRAD)]
rs = np.random.RandomState(0)


def dev(a):
    return Variable(gpu.to_device(np.ascontiguousarray(a.astype(np.float32))))


for B in (1, 2, 4, 8, 16):
    x = dev(rs.standard_normal((B, 256, T, 1)))
    c = dev(rs.standard_normal((B, 192, T, 1)))
    ps = [dev(rs.standard_normal((256, 256, 2, 1)) / 22), dev(np.zeros(256)), dev(rs.standard_normal((256, 192, 1, 1)) / 14), dev(np.zeros(256)),
          dev(rs.standard_normal((256, 128, 1, 1)) / 11), dev(np.zeros(256)), dev(rs.standard_normal((256, 128, 1, 1)) / 11), dev(np.zeros(256))]
    g1, g2 = dev(rs.standard_normal((B, 256, T, 1))), dev(rs.standard_normal((B, 256, T, 1)))

    def run():
        res, skip = ResidualBlockFunction(dil).apply([x, c] + ps)
        res.creator.backward(tuple(range(10)), (g1, g2))
    for _ in range(2):
        run()
    gpu.synchronize()
    lib.vqvae_prof_reset()
    lib.vqvae_prof_enable(sum(1 << t for _, t in TAGS))
    for _ in range(5):
        run()
    gpu.synchronize()
    lib.vqvae_prof_enable(0)
    out = []
    for name, t in TAGS:
        tot, cnt = C.c_double(0), C.c_int(0)
        _lib.call('vqvae_prof_read', t, C.byref(tot), C.byref(cnt))
        out.append('%s %7.1f us (x%d)' % (name, 1e3 * tot.value / max(cnt.value, 1), cnt.value // 5))
    print('B %2d  ' % B + ' | '.join(out))
