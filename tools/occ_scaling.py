"""Time of the dilated conv forward (conv_gemm_x3_kernel, 256 x 256 tiles forced with VQVAE_X3_NB=3) against the
number of workgroups in flight: B = 1, 2, 4, 8, 16 -> 30 ... 480 tiles on 256 CUs.  If a tile costs the same
whether 30 or 240 CUs are busy, the kernel is bound inside the CU (issue / latency); if it gets slower as the chip
fills, something shared (power, fabric) is the limit.
usage: VQVAE_X3_NB=3 python tools/occ_scaling.py"""
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
T = 7680


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
for dil in (64,):
    for B in (1, 2, 4, 8, 9, 16, 32):
        x = Variable(gpu.to_device(rs.standard_normal((B, 256, T, 1)).astype(np.float32)))
        W = Variable(gpu.to_device((rs.standard_normal((256, 256, 2, 1)) / 16).astype(np.float32)))
        b = Variable(gpu.to_device(rs.standard_normal(256).astype(np.float32)))
        with V.core.no_backprop_mode():
            us = timeit(_lib.PROF_CONV_FWD, lambda: F.convolution_1d(x, W, b, pad=dil, dilate=dil, out_len=T))
        tiles = B * 30
        rounds = -(-tiles // 256)
        print('B %2d  tiles %4d  rounds %d  %7.1f us  per round %6.1f us  %6.1f TFLOP/s' % (B, tiles, rounds, us, us / rounds, 2.0 * B * T * 256 * 512 / us / 1e6))
