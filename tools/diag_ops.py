"""Which operator of the device backward carries the 1e-4-level deviation seen in whole-step parity?
Compares single operators with the oracle evaluated in float64 (the fp32 oracle itself is 1e-6)."""
import sys
sys.path[:0] = ['tests', 'oracle', 'chainer-vq-vae_amd']
import numpy as np
import vqvae_oracle as O
from helpers import to4
from test_gpu_kernels import _rb_params
from vqvae_amd import backend as gpu, functions as F
from vqvae_amd.core import Variable
from vqvae_amd.wavenet import ResidualBlockFunction
gpu.init(0)


def rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return gpu.to_device(np.ascontiguousarray(a))


def d64(t):
    if isinstance(t, dict): return {k: d64(v) for k, v in t.items()}
    if isinstance(t, tuple): return tuple(d64(v) for v in t)
    return t.astype(np.float64) if isinstance(t, np.ndarray) and t.dtype == np.float32 else t


T = 7680
for dil in (1, 512):
    rs = np.random.RandomState(3 + dil)
    p = _rb_params(rs, 256, 256, 256, 192, 2)
    x = rs.standard_normal((1, 256, T)).astype(np.float32)
    c = rs.standard_normal((1, 192, T)).astype(np.float32)
    p64 = d64(p)
    res_ref, skip_ref, cache = O.resblock_fwd(p64, x.astype(np.float64), c.astype(np.float64), dil)
    g_res = rs.standard_normal(res_ref.shape).astype(np.float32)
    g_skip = rs.standard_normal(skip_ref.shape).astype(np.float32)
    gx_ref, gc_ref, gr = O.resblock_bwd(p64, cache, c.astype(np.float64), dil, g_res.astype(np.float64), g_skip.astype(np.float64))
    r32, s32, c32 = O.resblock_fwd(p, x, c, dil)
    gx32, gc32, gr32 = O.resblock_bwd(p, c32, c, dil, g_res, g_skip)
    order = ['conv', 'condition_proj', 'res', 'skip']
    vs = [Variable(dev(to4(x))), Variable(dev(to4(c)))]
    for n in order:
        vs += [Variable(dev(to4(p[n][0]))), Variable(dev(p[n][1]))]
    res, skip = ResidualBlockFunction(dil).apply(vs)
    fn = res.creator
    print('dil %d  fwd: res dev %.2e (oracle32 %.2e)  skip dev %.2e (%.2e)  gates: tanh %.2e sig %.2e z %.2e' % (
        dil, rel(res.data.get(), res_ref), rel(r32, res_ref), rel(skip.data.get(), skip_ref), rel(s32, skip_ref),
        rel(fn.gates.get()[:, :128], cache[1]), rel(fn.gates.get()[:, 128:], cache[2]), rel(fn.z.get(), cache[3])))
    gouts = fn.backward(tuple(range(10)), (Variable(dev(to4(g_res))), Variable(dev(to4(g_skip)))))
    print('   bwd: gx dev %.2e (oracle32 %.2e)  gcond %.2e (%.2e)' % (rel(gouts[0].get(), gx_ref), rel(gx32, gx_ref), rel(gouts[1].get(), gc_ref), rel(gc32, gc_ref)))
    for i, n in enumerate(order):
        print('        gW %-14s dev %.2e (oracle32 %.2e)   gb dev %.2e (%.2e)' % (n, rel(gouts[2 + 2 * i].get(), gr[n][0]), rel(gr32[n][0], gr[n][0]), rel(gouts[3 + 2 * i].get(), gr[n][1]), rel(gr32[n][1], gr[n][1])))

# plain 1x1 conv fwd / bwd-data / bwd-weight at K = 256 and softmax-CE backward
rs = np.random.RandomState(5)
x = rs.standard_normal((1, 256, T, 1)).astype(np.float32)
W = (rs.standard_normal((256, 256, 1, 1)) / 16).astype(np.float32)
b = rs.standard_normal(256).astype(np.float32)
gy = rs.standard_normal((1, 256, T, 1)).astype(np.float32)
vx, vW, vb = Variable(dev(x)), Variable(dev(W)), Variable(dev(b))
y = F.convolution_1d(vx, vW, vb)
y.grad = dev(gy); y.backward()
x64, W64, g64 = x[..., 0].astype(np.float64), W[..., 0].astype(np.float64), gy[..., 0].astype(np.float64)
y64 = O.conv1d_fwd(x64, W64, b.astype(np.float64))
gx64, gW64, gb64 = O.conv1d_bwd(x64, W64, g64)
print('1x1 conv K=256: fwd %.2e  gx %.2e  gW %.2e  gb %.2e' % (rel(y.data.get(), y64), rel(vx.grad.get(), gx64), rel(vW.grad.get(), gW64), rel(vb.grad.get(), gb64)))
yl = (3 * rs.standard_normal((1, 256, T, 1))).astype(np.float32)
tg = rs.randint(0, 256, size=(1, T, 1)).astype(np.int32)
vy = Variable(dev(yl))
loss = F.softmax_cross_entropy(vy, Variable(dev(tg)))
loss.backward()
l64, logp = O.softmax_xent_fwd(yl[..., 0].astype(np.float64), tg[..., 0])
print('softmax-CE: loss %.2e  gy %.2e' % (abs(float(loss.data.get()) - l64) / l64, rel(vy.grad.get(), O.softmax_xent_bwd(logp, tg[..., 0]))))
