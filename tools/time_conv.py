"""Times single conv1d entry points at configs[1] shapes with the library's HIP-event profiler.
usage: python tools/time_conv.py"""
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
B, T = 16, 7680


def timeit(tag, fn, n=20):
    for _ in range(3):
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
for (Cin, Cout, K, dil, name) in [(128, 256, 1, 1, 'res 1x1 (K=128 -> 256 rows)'), (256, 256, 1, 1, '1x1 256->256'),
                                  (256, 256, 2, 64, 'dilated k=2 256->256'), (2560, 256, 1, 1, 'skip sum K=2560')]:
    x = Variable(gpu.to_device(rs.standard_normal((B, Cin, T, 1)).astype(np.float32)))
    W = Variable(gpu.to_device((rs.standard_normal((Cout, Cin, K, 1)) / np.sqrt(Cin * K)).astype(np.float32)))
    b = Variable(gpu.to_device(rs.standard_normal(Cout).astype(np.float32)))
    with V.core.no_backprop_mode():
        us = timeit(_lib.PROF_CONV_FWD, lambda: F.convolution_1d(x, W, b, pad=(K - 1) * dil, dilate=dil, out_len=T))
    flop = 2.0 * B * T * Cin * Cout * K
    byt = 4.0 * B * T * (Cin + Cout)
    print('%-30s %7.1f us  %6.1f TFLOP/s  %5.2f TB/s (x in + y out)' % (name, us, flop / us / 1e6, byt / us / 1e6))
