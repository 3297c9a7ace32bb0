"""Accuracy of the matmul modes against float64 on the conv1d entry points (fwd, bwd-data, bwd-weight)."""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'chainer-vq-vae_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from vqvae_amd import backend as gpu, functions as F
from vqvae_amd.core import Variable


def conv64(x, W, b, stride, pad, dil, crop):
    B, Cin, Tin = x.shape
    Cout, _, K = W.shape
    nat = (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1
    xp = np.zeros((B, Cin, Tin + 2 * pad), np.float64)
    xp[:, :, pad:pad + Tin] = x
    y = np.zeros((B, Cout, nat), np.float64)
    for k in range(K):
        xs = xp[:, :, k * dil: k * dil + (nat - 1) * stride + 1: stride]
        y += np.einsum('oc,bct->bot', W[:, :, k].astype(np.float64), xs)
    y += b.astype(np.float64)[None, :, None]
    return y if crop is None else y[:, :, :crop]


def bwd64(x, W, gy, stride, pad, dil):
    B, Cin, Tin = x.shape
    Cout, _, K = W.shape
    nat = gy.shape[2]
    xp = np.zeros((B, Cin, Tin + 2 * pad), np.float64)
    xp[:, :, pad:pad + Tin] = x
    gxp = np.zeros_like(xp)
    gW = np.zeros(W.shape, np.float64)
    for k in range(K):
        sl = slice(k * dil, k * dil + (nat - 1) * stride + 1, stride)
        gW[:, :, k] = np.einsum('bot,bct->oc', gy.astype(np.float64), xp[:, :, sl])
        gxp[:, :, sl] += np.einsum('oc,bot->bct', W[:, :, k].astype(np.float64), gy.astype(np.float64))
    return gxp[:, :, pad:pad + Tin], gW


CASES = [
    (2, 32, 256, 32, 4, 2, 1, 1, None), (2, 96, 300, 80, 2, 1, 8, 8, 300), (2, 256, 384, 64, 2, 1, 1, 1, 384),
    (2, 48, 200, 30, 1, 1, 0, 1, None), (2, 40, 77, 50, 3, 2, 2, 3, None), (2, 1280, 120, 192, 1, 1, 0, 1, None),
    (2, 192, 120, 1280, 1, 1, 0, 1, None), (3, 520, 90, 70, 3, 1, 2, 2, None), (1, 256, 1000, 512, 2, 1, 3, 3, 1000),
    (2, 256, 2048, 256, 2, 1, 64, 64, 2048), (2, 512, 1024, 512, 1, 1, 0, 1, None),
]
for case in CASES:
    B, Cin, Tin, Cout, K, stride, pad, dil, crop = case
    rs = np.random.RandomState(zlib.crc32(repr(case).encode()))
    x = rs.standard_normal((B, Cin, Tin)).astype(np.float32)
    W = (rs.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    y64 = conv64(x, W, b, stride, pad, dil, crop)
    gy = rs.standard_normal(y64.shape).astype(np.float32)
    nat = (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1
    gfull = np.zeros((B, Cout, nat), np.float32)
    gfull[:, :, :gy.shape[2]] = gy
    gx64, gW64 = bwd64(x, W, gfull, stride, pad, dil)
    line = '%-44s' % (case,)
    for mode in ('float32', 'float32x3'):
        gpu.set_matmul_dtype(mode)
        vx = Variable(gpu.to_device(x[..., None])); vW = Variable(gpu.to_device(W[..., None])); vb = Variable(gpu.to_device(b))
        y = F.convolution_1d(vx, vW, vb, stride=stride, pad=pad, dilate=dil, out_len=crop)
        ey = np.abs(y.data.get()[..., 0] - y64).max() / np.abs(y64).max()
        y.grad = gpu.to_device(gy[..., None]); y.backward()
        ex = np.abs(vx.grad.get()[..., 0] - gx64).max() / np.abs(gx64).max()
        ew = np.abs(vW.grad.get()[..., 0] - gW64).max() / np.abs(gW64).max()
        line += ' | %-9s y %.2e gx %.2e gW %.2e' % (mode, ey, ex, ew)
    print(line)
