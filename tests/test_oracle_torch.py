"""Independent second opinion on the oracle's Chainer-recalled parts (SURVEY 8c "how the unpinned
parts get pinned anyway", item 5): conv / dilated conv forward and gradients, align-corners
up-sampling, softmax cross-entropy, the gated residual block and the Adam step, against PyTorch's
CPU implementations of the same published operators.  Dev/CI container only -- torch is test
tooling here; nothing in the package imports it."""
import numpy as np
import pytest

import vqvae_oracle as O

torch = pytest.importorskip('torch')
F = torch.nn.functional


def _t(a, grad=False):
    return torch.tensor(np.asarray(a, np.float64), requires_grad=grad)


@pytest.mark.parametrize('stride,pad,dil,K', [(2, 1, 1, 4), (1, 4, 4, 3), (1, 16, 16, 2), (1, 0, 1, 1), (2, 2, 3, 3)])
def test_conv1d_matches_torch(stride, pad, dil, K):
    rs = np.random.RandomState(K * 7 + dil)
    x = rs.standard_normal((2, 5, 41))
    W = rs.standard_normal((6, 5, K))
    b = rs.standard_normal(6)
    y = O.conv1d_fwd(x, W, b, stride, pad, dil)
    tx, tW, tb = _t(x, True), _t(W, True), _t(b, True)
    ty = F.conv1d(tx, tW, tb, stride=stride, padding=pad, dilation=dil)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-12, atol=1e-12)
    gy = rs.standard_normal(y.shape)
    ty.backward(_t(gy))
    gx, gW, gb = O.conv1d_bwd(x, W, gy, stride, pad, dil)
    np.testing.assert_allclose(gx, tx.grad.numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gW, tW.grad.numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gb, tb.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_causal_dilated_conv_is_left_padded_conv():
    """modules.py:13-16, 40-41: pad = dil on both sides then crop [:T] == left-pad only."""
    rs = np.random.RandomState(1)
    x = rs.standard_normal((2, 4, 50))
    W = rs.standard_normal((8, 4, 2))
    b = rs.standard_normal(8)
    for dil in (1, 2, 8):
        want = F.conv1d(F.pad(_t(x), (dil, 0)), _t(W), _t(b), dilation=dil).numpy()
        np.testing.assert_allclose(O.causal_conv_fwd(x, W, b, dil), want, rtol=1e-12, atol=1e-12)


def test_upsample_matches_torch_align_corners():
    """F.resize_images (net.py:54, 60) on a (T', 1) image == linear interpolation with
    align_corners=True, forward and backward."""
    rs = np.random.RandomState(2)
    x = rs.standard_normal((2, 3, 120))
    tx = _t(x, True)
    ty = F.interpolate(tx, size=7680, mode='linear', align_corners=True)
    np.testing.assert_allclose(O.upsample_fwd(x, 7680), ty.detach().numpy(), rtol=1e-6, atol=1e-6)
    gy = rs.standard_normal((2, 3, 7680))
    ty.backward(_t(gy))
    np.testing.assert_allclose(O.upsample_bwd(gy, 120), tx.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_softmax_xent_matches_torch():
    rs = np.random.RandomState(3)
    y = rs.standard_normal((3, 11, 17)) * 3
    t = rs.randint(0, 11, (3, 17)).astype(np.int32)
    loss, logp = O.softmax_xent_fwd(y, t)
    ty = _t(y, True)
    tl = F.cross_entropy(ty, torch.tensor(t.astype(np.int64)))          # mean over B*T, class axis 1
    assert abs(float(loss) - float(tl.detach())) < 1e-12
    tl.backward()
    np.testing.assert_allclose(O.softmax_xent_bwd(logp, t), ty.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_resblock_and_wavenet_gradients_match_torch_autograd():
    """modules.py:30-56, 89-96, 148-160 composed from torch ops; the oracle's hand-written backward
    against autograd (float64)."""
    rs = np.random.RandomState(4)
    cfg = dict(n_loop=2, n_layer=3, residual=6, dilated=8, skip=5, input_dim=7, out_dim=7,
               local_dim=2, global_dim=1, d=2, k=4, n_speaker=2)
    p = O.make_params(rs, dtype=np.float64, **cfg)['decoder']
    for blk in p['blocks']:
        for name in blk:
            blk[name] = (blk[name][0], 0.1 * rs.standard_normal(blk[name][1].shape))
    B, T = 2, 19
    x = rs.standard_normal((B, 7, T))
    cond = rs.standard_normal((B, 3, T))
    y, cache = O.wavenet_fwd(p, x, cond, 2, 3)
    gy = rs.standard_normal(y.shape)
    gcond, G = O.wavenet_bwd(p, cache, cond, gy, 2, 3)

    leaves = {}

    def leaf(name, arr):
        leaves[name] = _t(arr, True)
        return leaves[name]
    tc = leaf('cond', cond)
    h = F.conv1d(F.pad(_t(x), (1, 0)), leaf('embed/W', p['embed'][0]), leaf('embed/b', p['embed'][1]))
    skip_sum = None
    for i, (blk, dil) in enumerate(zip(p['blocks'], O.wavenet_dilations(2, 3))):
        g = F.conv1d(F.pad(h, (dil, 0)), leaf('%d/conv/W' % i, blk['conv'][0]), leaf('%d/conv/b' % i, blk['conv'][1]), dilation=dil)
        g = g + F.conv1d(tc, leaf('%d/cp/W' % i, blk['condition_proj'][0]), leaf('%d/cp/b' % i, blk['condition_proj'][1]))
        a, s = torch.chunk(g, 2, dim=1)
        z = torch.tanh(a) * torch.sigmoid(s)
        sk = F.conv1d(z, leaf('%d/skip/W' % i, blk['skip'][0]), leaf('%d/skip/b' % i, blk['skip'][1]))
        h = F.conv1d(z, leaf('%d/res/W' % i, blk['res'][0]), leaf('%d/res/b' % i, blk['res'][1])) + h
        skip_sum = sk if skip_sum is None else skip_sum + sk
    z1 = torch.relu(F.conv1d(torch.relu(skip_sum), leaf('proj1/W', p['proj1'][0]), leaf('proj1/b', p['proj1'][1])))
    ty = F.conv1d(z1, leaf('proj2/W', p['proj2'][0]), leaf('proj2/b', p['proj2'][1]))
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-10, atol=1e-10)
    ty.backward(_t(gy))
    np.testing.assert_allclose(gcond, leaves['cond'].grad.numpy(), rtol=1e-9, atol=1e-10)
    for name in ('embed', 'proj1', 'proj2'):
        np.testing.assert_allclose(G[name][0], leaves[name + '/W'].grad.numpy(), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(G[name][1], leaves[name + '/b'].grad.numpy(), rtol=1e-9, atol=1e-10)
    last = len(p['blocks']) - 1
    for i, bg in enumerate(G['blocks']):
        for oname, tname in (('conv', 'conv'), ('condition_proj', 'cp'), ('skip', 'skip'), ('res', 'res')):
            if oname == 'res' and i == last:
                # the last block's residual output is unused (modules.py:92-96): no gradient reaches it
                assert leaves['%d/res/W' % i].grad is None and (bg['res'] is None or bg['res'][0] is None or not np.any(bg['res'][0]))
                continue
            np.testing.assert_allclose(bg[oname][0], leaves['%d/%s/W' % (i, tname)].grad.numpy(), rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(bg[oname][1], leaves['%d/%s/b' % (i, tname)].grad.numpy(), rtol=1e-9, atol=1e-10)


def test_adam_step_matches_torch_when_eps_is_negligible():
    """chainer Adam: p -= alpha*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps); torch puts eps on
    sqrt(v_hat) instead -- identical up to eps*sqrt(1-b2^t) in the denominator."""
    rs = np.random.RandomState(5)
    p0 = rs.standard_normal(50)
    tp = torch.nn.Parameter(_t(p0))
    opt = torch.optim.Adam([tp], lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    p, m, v = p0.copy(), np.zeros(50), np.zeros(50)
    for t in range(1, 4):
        g = rs.standard_normal(50)
        g = np.sign(g) * (1 + np.abs(g))          # |g| >= 1: eps (1e-8) is negligible in both placements
        tp.grad = _t(g)
        opt.step()
        O.adam_update(p, g, m, v, t, 2e-4)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=0, atol=1e-9)
