"""Weight gradient of stride-2 convs (vqvae_conv1d_bwd_weight) against a float64 einsum over a grid of shapes (dev check)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'chainer-vq-vae_amd')]
from vqvae_amd import backend as gpu, functions as F
from vqvae_amd.core import Variable
gpu.init(0)
rs = np.random.RandomState(0)
bad = 0
ONLY = int(sys.argv[1]) if len(sys.argv) > 1 else -1
case = -1
for (B, Cin, Cout) in ((2, 16, 32), (3, 64, 256), (1, 256, 256)):
    for Tin in (8, 12, 16, 24, 30, 32, 60, 96, 120, 160, 240, 250, 480):
        case += 1
        if ONLY >= 0 and case != ONLY:
            continue
        print('case', case, B, Cin, Cout, Tin, flush=True)
        K, stride, pad = 4, 2, 1
        Tout = (Tin + 2 * pad - K) // stride + 1
        x = rs.standard_normal((B, Cin, Tin)).astype(np.float32)
        W = (rs.standard_normal((Cout, Cin, K)) / 8).astype(np.float32)
        gy = rs.standard_normal((B, Cout, Tout)).astype(np.float32)
        vx = Variable(gpu.to_device(x[..., None])); vW = Variable(gpu.to_device(W[..., None]))
        y = F.convolution_1d(vx, vW, None, stride=stride, pad=pad)
        assert y.shape[2] == Tout, (y.shape, Tout)
        y.grad = gpu.to_device(gy[..., None])
        y.backward()
        xp = np.zeros((B, Cin, Tin + 2 * pad + 4), np.float64); xp[:, :, pad:pad + Tin] = x
        want = np.zeros((Cout, Cin, K))
        for j in range(K):
            want[:, :, j] = np.einsum('bot,bit->oi', gy.astype(np.float64), xp[:, :, j:j + stride * Tout:stride])
        got = vW.grad.get()[..., 0]
        err = np.abs(got - want).max() / np.abs(want).max()
        flag = '' if err < 1e-4 else '  <-- BAD'
        bad += err >= 1e-4
        print('B %d Cin %3d Cout %3d Tin %3d Tout %3d  err %.2e%s' % (B, Cin, Cout, Tin, Tout, err, flag))
print('bad', bad)
