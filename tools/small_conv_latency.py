"""How long a latent-rate conv launch takes as a function of its K steps (B = 16, T' = 120, 64 output channels, k = 3:
16 workgroups): the fixed cost and the slope per 16-channel step, per matmul mode.  usage: python tools/small_conv_latency.py"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'chainer-vq-vae_amd')]
from vqvae_amd import _lib, backend as gpu, functions as F
from vqvae_amd.core import Variable
import vqvae_amd as V

gpu.init(0)
lib = _lib.load()
B, T = 16, 120


def timeit(tag, fn, n=50):
    for _ in range(5):
        fn()
    gpu.synchronize()
    lib.vqvae_prof_reset(); lib.vqvae_prof_enable(1 << tag)
    for _ in range(n):
        fn()
    gpu.synchronize()
    lib.vqvae_prof_enable(0)
    tot, cnt = C.c_double(0), C.c_int(0)
    _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
    return 1e3 * tot.value / max(cnt.value, 1)


rs = np.random.RandomState(0)
for mode in ('float32x3', 'bfloat16', 'float32'):
    gpu.set_matmul_dtype(mode)
    row = []
    for Cin in (16, 64, 256, 1024):
        x = Variable(gpu.to_device(rs.standard_normal((B, Cin, T, 1)).astype(np.float32)))
        W = Variable(gpu.to_device((rs.standard_normal((64, Cin, 3, 1)) / np.sqrt(Cin * 3)).astype(np.float32)))
        b = Variable(gpu.to_device(rs.standard_normal(64).astype(np.float32)))
        with V.core.no_backprop_mode():
            us = timeit(_lib.PROF_CONV_FWD, lambda: F.convolution_1d(x, W, b, pad=1, dilate=1, out_len=T))
        row.append('%d steps: %.1f us' % (3 * Cin // 16, us))
    print(mode, ' | '.join(row))
