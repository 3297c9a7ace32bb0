"""Slot timing of the PING-PONG EXPERIMENT's conv kernel (not the product): apply
tools/experiments/conv_gemm_pingpong.patch to conv_gemm.hip of commit b3f89f9, build with `make EXTRA=-DX3_PROF`,
then run this: it prints the shader-clock stamps that workgroup 0 recorded at the barriers of K steps 8..15 of one
dilated-conv launch (fetch / fragment reads / stage / barrier / MFMA issue / barrier, all eight waves).
usage: VQVAE_X3_NB=3 python tools/pp_prof.py [B]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T, dil = 7680, 64
rs = np.random.RandomState(0)
x = Variable(gpu.to_device(rs.standard_normal((B, 256, T, 1)).astype(np.float32)))
W = Variable(gpu.to_device((rs.standard_normal((256, 256, 2, 1)) / 16).astype(np.float32)))
b = Variable(gpu.to_device(rs.standard_normal(256).astype(np.float32)))
with V.core.no_backprop_mode():
    for _ in range(3):
        F.convolution_1d(x, W, b, pad=dil, dilate=dil, out_len=T)
gpu.synchronize()
buf = (C.c_ulonglong * 512)()
assert lib.vqvae_debug_x3_prof(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(8, 4, 16)
base = st[0, 0, 0]
print('absolute stamps (cycles since wave 0 entered step 10), steps 10 (even) and 11 (odd), all waves:')
print('wave |   top  fetch  frags  stage | sync1   mfma | sync2=top fetch  frags  stage | sync1   mfma | sync2')
base = st[0, 1, 0]
for w in range(8):
    r = st[w, 1] - base
    print('  %d  | %5d %6d %6d %6d | %5d %6d | %5d %6d %6d %6d | %5d %6d | %5d' % (w, r[0], r[1], r[2], r[3], r[4], r[5], r[8], r[9], r[10], r[11], r[12], r[13], r[14]))
