"""CPU tests of the oracle itself (-m "not gpu"): golden vectors captured from the
reference's own utils.py, analytic known-answer tests for the Chainer semantics
that nothing in the reference pins, and fp64 finite-difference gradient checks
of every backward."""
import os

import numpy as np
import pytest

import vqvae_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# ---- golden vectors (reference utils.py executed in the dev container) --------
def test_mulaw_golden():
    g = np.load(os.path.join(GOLD, 'mulaw.npz'))
    mu = O.MuLaw(256)
    np.testing.assert_array_equal(mu.transform(g['x']), g['q'])
    np.testing.assert_array_equal(mu.itransform(np.arange(256)), g['itransform'])
    # SURVEY 8c probe: [-1,-.5,-1e-4,0,1e-4,.5,1] -> [0,15,127,128,128,240,255]
    np.testing.assert_array_equal(g['q'][:7], [0, 15, 127, 128, 128, 240, 255])


@pytest.mark.parametrize('name', ['vq_train', 'vq_3d', 'vq_ties'])
def test_vq_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    e, idx = O.vq_forward(g['z'], g['W'])
    np.testing.assert_array_equal(idx, g['idx'])
    np.testing.assert_array_equal(np.ascontiguousarray(e), g['e'])
    assert not e.flags['C_CONTIGUOUS'] or e.ndim < 3     # same transposed view as utils.py:206-210
    gy = g['gy']
    gx, gW = O.vq_backward(idx, g['W'], gy)
    assert gx is gy                                      # identity, same object (utils.py:218-219)
    np.testing.assert_array_equal(gW, g['gW'])


def test_vq_golden_ties_first_index_wins():
    g = np.load(os.path.join(GOLD, 'vq_ties.npz'))
    assert g['idx'][0, 0, 0] == 7      # W[7] == W[40] == W[77]: first index
    assert g['idx'][0, 1, 0] == 3      # W[3] == W[90]


def test_vq_golden_stress():
    from golden.make_golden import stress_inputs
    g = np.load(os.path.join(GOLD, 'vq_stress.npz'))
    B, d, T, k = [int(v) for v in g['shape']]
    z, W = stress_inputs(int(g['seed_z']), int(g['seed_w']), B, d, T, k)
    e, idx = O.vq_forward_chunked(z, W, 1)
    np.testing.assert_array_equal(idx, g['idx'])
    gy = np.random.RandomState(int(g['seed_gy'])).standard_normal((B, d, T, 1)).astype(np.float32)
    _, gW = O.vq_backward(idx, W, gy)
    np.testing.assert_array_equal(gW[g['gW_rows']], g['gW_vals'])


def test_vq_sequential_sum_property():
    """The reference's distance equals a sequential-over-d fp32 accumulation bit for
    bit (what the exact HIP kernel evaluates)."""
    g = np.load(os.path.join(GOLD, 'vq_train.npz'))
    z = g['z'].reshape(g['z'].shape[:3])
    W = g['W']
    acc = np.zeros((z.shape[0], W.shape[0], z.shape[2]), np.float32)
    for c in range(z.shape[1]):
        df = z[:, None, c, :] - W[None, :, c, None]
        acc = acc + df * df
    ref = np.sum((z[:, None] - W[None, :, :, None]) ** 2, axis=2)
    np.testing.assert_array_equal(acc, ref)


# ---- known-answer tests for the [chainer-recalled] semantics --------------------
def test_conv_impulse_reads_back_taps():
    x = np.zeros((1, 1, 32), np.float32)
    x[0, 0, 10] = 1
    W = np.array([[[2.0, 3.0]]], np.float32)
    y = O.causal_conv_fwd(x, W, None, 4)
    want = np.zeros(32, np.float32)
    want[10], want[14] = 3, 2              # tap1 <-> x[t], tap0 <-> x[t-dil]
    np.testing.assert_array_equal(y[0, 0], want)
    ye = O.conv1d_fwd(x, W, None, 1, 1, 1)[:, :, :32]   # embed conv: y[t] = E0 x[t-1] + E1 x[t]
    want = np.zeros(32, np.float32)
    want[10], want[11] = 3, 2
    np.testing.assert_array_equal(ye[0, 0], want)


def test_encoder_lengths():
    L = 7681
    for _ in range(6):
        L = O.conv_out_len(L, 4, 2, 1, 1)
    assert L == 120                         # 7681 -> 120, 120*64 = 7680


def test_resize_align_corners():
    v0, v1, w0, w1 = O.resize_tables(120, 7680)
    assert v0[0] == 0 and w0[0] == 1 and w1[0] == 0
    assert v0[-1] == 118 and v1[-1] == 119 and w0[-1] == 0 and w1[-1] == 1
    x = np.arange(120, dtype=np.float32)[None, None]
    y = O.upsample_fwd(x, 7680)
    np.testing.assert_allclose(y[0, 0], np.arange(7680) * 119 / 7679, rtol=1e-5, atol=1e-4)
    c = np.full((1, 1, 9), 2.5, np.float32)
    np.testing.assert_allclose(O.upsample_fwd(c, 576), 2.5, rtol=1e-6)
    v0, v1, w0, w1 = O.resize_tables(1, 7)           # degenerate axis = broadcast
    assert (w0 == 0).all() and (w1 == 1).all() and (v1 == 0).all()


def test_softmax_xent_uniform_is_ln_q():
    y = np.zeros((2, 256, 50), np.float32)
    t = np.random.RandomState(0).randint(0, 256, (2, 50)).astype(np.int32)
    loss, _ = O.softmax_xent_fwd(y, t)
    assert abs(float(loss) - np.log(256.0)) < 1e-6   # loss1.png starts at ~5.5


def test_adam_first_step_closed_form():
    rs = np.random.RandomState(0)
    p = rs.standard_normal(1000)
    g = rs.standard_normal(1000)
    p0 = p.copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    O.adam_update(p, g, m, v, 1, 2e-4)
    want = p0 - 2e-4 * g / (np.abs(g) + 1e-8 / np.sqrt(1 - 0.999))
    np.testing.assert_allclose(p, want, rtol=1e-9, atol=1e-12)


def test_ema_weights_are_swapped():
    e = np.ones(4, np.float32)
    t = np.zeros(4, np.float32)
    O.ema_update(e, t, 0.9999)            # utils.py:153-154: 0.9999*target + 0.0001*ema
    np.testing.assert_allclose(e, 1e-4, rtol=1e-3)


def test_loss3_is_beta_loss2():
    rs = np.random.RandomState(0)
    cfg = dict(d=8, k=16, n_loop=1, n_layer=2, residual=16, dilated=32, skip=16, out_dim=256,
               local_dim=8, global_dim=8, n_speaker=3)
    P = O.make_params(rs, **cfg)
    b = O.synth_batch(1, length=128, n_speaker=3, seed=1)
    (l1, l2, l3), _ = O.vae_forward(P, *b, 1, 2)
    assert abs(float(l3) - 0.25 * float(l2)) < 1e-7      # loss2.png/loss3.png: 48 <-> 12


# ---- fp64 finite-difference gradient checks ------------------------------------
def _fd(f, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=['multi_index'])
    while not it.finished:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        fp = f()
        x[i] = old - eps
        fm = f()
        x[i] = old
        g[i] = (fp - fm) / (2 * eps)
        it.iternext()
    return g


@pytest.mark.parametrize('stride,pad,dil,K', [(2, 1, 1, 4), (1, 4, 4, 3), (1, 2, 2, 2), (1, 0, 1, 1)])
def test_conv_grad_fd(stride, pad, dil, K):
    rs = np.random.RandomState(1)
    x = rs.standard_normal((2, 3, 17))
    W = rs.standard_normal((4, 3, K))
    b = rs.standard_normal(4)
    r = rs.standard_normal(O.conv1d_fwd(x, W, b, stride, pad, dil).shape)
    f = lambda: float((O.conv1d_fwd(x, W, b, stride, pad, dil) * r).sum())
    gx, gW, gb = O.conv1d_bwd(x, W, r, stride, pad, dil)
    np.testing.assert_allclose(gx, _fd(f, x), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gW, _fd(f, W), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gb, _fd(f, b), rtol=1e-6, atol=1e-7)


def test_resblock_grad_fd():
    rs = np.random.RandomState(2)
    Cr, Cd, Cs, Cc, T, dil = 4, 6, 5, 3, 12, 2
    def conv(co, ci, k):
        return (rs.standard_normal((co, ci, k)) * 0.5, rs.standard_normal(co) * 0.1)
    p = {'conv': conv(Cd, Cr, 2), 'condition_proj': conv(Cd, Cc, 1), 'res': conv(Cr, Cd // 2, 1),
         'skip': conv(Cs, Cd // 2, 1)}
    x = rs.standard_normal((2, Cr, T))
    c = rs.standard_normal((2, Cc, T))
    r1 = rs.standard_normal((2, Cr, T))
    r2 = rs.standard_normal((2, Cs, T))

    def f():
        res, skip, _ = O.resblock_fwd(p, x, c, dil)
        return float((res * r1).sum() + (skip * r2).sum())
    _, _, cache = O.resblock_fwd(p, x, c, dil)
    gx, gc, gr = O.resblock_bwd(p, cache, c, dil, r1, r2)
    np.testing.assert_allclose(gx, _fd(f, x), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gc, _fd(f, c), rtol=1e-5, atol=1e-7)
    for n in p:
        np.testing.assert_allclose(gr[n][0], _fd(f, p[n][0]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gr[n][1], _fd(f, p[n][1]), rtol=1e-5, atol=1e-7)


def test_upsample_and_xent_grad_fd():
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1, 2, 5))
    r = rs.standard_normal((1, 2, 40))
    f = lambda: float((O.upsample_fwd(x, 40) * r).sum())
    np.testing.assert_allclose(O.upsample_bwd(r, 5), _fd(f, x), rtol=1e-6, atol=1e-8)
    y = rs.standard_normal((2, 7, 6))
    t = rs.randint(0, 7, (2, 6)).astype(np.int32)
    f = lambda: float(O.softmax_xent_fwd(y, t)[0])
    _, logp = O.softmax_xent_fwd(y, t)
    np.testing.assert_allclose(O.softmax_xent_bwd(logp, t), _fd(f, y), rtol=1e-5, atol=1e-8)


def test_full_model_grad_fd_spot():
    """Spot-check the three-loss gradient of the whole model in fp64 (decoder and
    condition-embed parameters; the VQ argmin is piecewise constant)."""
    rs = np.random.RandomState(4)
    cfg = dict(d=4, k=6, n_loop=1, n_layer=2, residual=4, dilated=4, skip=4, out_dim=8, input_dim=8,
               local_dim=4, global_dim=3, n_speaker=2, dtype=np.float64)
    P = O.make_params(rs, **cfg)
    L = 128
    x_enc = rs.standard_normal((1, 1, L + 1))
    q = rs.randint(0, 8, (1, L + 1))
    x_dec = np.identity(8)[q[:, :-1]].transpose(0, 2, 1).copy()
    t = q[:, 1:].astype(np.int32)
    spk = np.array([1], np.int32)
    (l1, l2, l3), cache = O.vae_forward(P, x_enc, x_dec, spk, t, 1, 2)
    G = O.vae_backward(P, cache, spk, t, 1, 2)
    idx0 = cache['idx'].copy()

    def loss1():
        (a, _, _), c = O.vae_forward(P, x_enc, x_dec, spk, t, 1, 2)
        assert (c['idx'] == idx0).all()
        return float(a)
    W = P['decoder']['blocks'][0]['conv'][0]
    g = _fd(loss1, W, 1e-6)
    np.testing.assert_allclose(G['decoder']['blocks'][0]['conv'][0], g, rtol=1e-4, atol=1e-8)
    E = P['condition_embed']['global_embed']
    np.testing.assert_allclose(G['condition_embed']['global_embed'], _fd(loss1, E, 1e-6), rtol=1e-4, atol=1e-8)
    Wl = P['condition_embed']['local_embed5'][0]
    np.testing.assert_allclose(G['condition_embed']['local_embed5'][0], _fd(loss1, Wl, 1e-6), rtol=1e-4, atol=1e-8)
    # last block's residual conv gets no gradient (modules.py:89-96)
    assert G['decoder']['blocks'][-1]['res'] is None


def test_mol_loss_runs_and_matches_bruteforce():
    rs = np.random.RandomState(5)
    y = rs.standard_normal((2, 30, 16)).astype(np.float32)
    t = rs.uniform(-1, 1, (2, 1, 16)).astype(np.float32)
    t[0, 0, 0], t[0, 0, 1] = -1.0, 1.0           # both edge branches (modules.py:198-209)
    loss = O.mol_loss_fwd(y, t)
    assert np.isfinite(loss)
    # brute force in float64 for one interior position
    b, i = 1, 5
    yy = y[b, :, i].astype(np.float64)
    lp, mu, ls = yy[:10], yy[10:20], np.maximum(yy[20:], -40)
    tt = 127.5 * float(t[b, 0, i])
    sig = lambda v: 1 / (1 + np.exp(-v))
    cdf = sig((tt - mu + 0.5) * np.exp(-ls)) - sig((tt - mu - 0.5) * np.exp(-ls))
    ref = -np.log(np.sum(np.exp(lp - np.log(np.exp(lp).sum())) * np.maximum(cdf, 1e-12)))
    full = O.mol_loss_fwd(y[b:b + 1, :, i:i + 1], t[b:b + 1, :, i:i + 1])
    assert abs(float(full) - ref) < 1e-4 * max(1, abs(ref))


def test_mol_grad_fd():
    rs = np.random.RandomState(6)
    t = rs.uniform(-0.9, 0.9, (2, 1, 5))
    t[0, 0, 1], t[1, 0, 2] = -1.0, 1.0              # both edge branches
    y = rs.standard_normal((2, 9, 5))
    y[:, 3:6] = 127.5 * t + 20 * rs.standard_normal((2, 3, 5))     # means near the targets
    y[:, 6:] = rs.uniform(1.0, 3.0, (2, 3, 5))                       # moderate scales: no saturation
    y[0, 6, 0] = -45.0                              # below log_scale_min: no gradient through the clamp
    f = lambda: float(O.mol_loss_fwd(y, t))
    g = O.mol_loss_bwd(y, t)
    np.testing.assert_allclose(g, _fd(f, y, 1e-6), rtol=2e-5, atol=1e-9)
    assert g[0, 6, 0] == 0.0


def test_preprocess_contract_golden():
    """Input contract of the hot path, pinned by running the reference's Preprocess.__call__
    (utils.py:54-110, librosa stubbed) on synthetic waveforms: padding and cropping branches,
    mu-law/one-hot and raw/logistic variants, shapes, dtypes and values."""
    g = np.load(os.path.join(GOLD, 'preprocess.npz'))
    for tag, mu_in, logi in (('mulaw', True, False), ('logistic', False, True)):
        for name, spk in (('short', 1), ('long', 2)):          # sorted speakers p225,p226,p227
            raw, x_dec, speaker, t = O.preprocess_contract(
                g['wave_' + name], 255, 256, int(g['crop_start']), mu_in, logi, spk)
            key = '%s_%s_' % (tag, name)
            for got, k in ((raw, 'raw'), (x_dec, 'x_dec'), (speaker, 'speaker'), (t, 't')):
                want = g[key + k]
                assert got.shape == want.shape and got.dtype == want.dtype, (key + k, got.shape, want.shape, got.dtype, want.dtype)
                np.testing.assert_array_equal(got, want)
    assert g['mulaw_short_x_dec'].shape == (256, 255, 1) and g['mulaw_short_t'].shape == (255, 1)
    assert g['mulaw_short_t'][-1, 0] == 128          # padded with bin quantize//2 (utils.py:74)


# --------------------------------------------------------------------------- #
# incremental generation
# --------------------------------------------------------------------------- #
GEN = dict(n_loop=2, n_layer=3, residual=16, dilated=16, skip=16, input_dim=12, out_dim=12,
           local_dim=4, global_dim=4, d=4, k=8, n_speaker=3)


def _gen_setup(T=24, n=2, seed=3, **over):
    cfg = dict(GEN, **over)
    rs = np.random.RandomState(seed)
    p = O.make_params(rs, **cfg)['decoder']
    for name in ('embed', 'proj1', 'proj2'):
        p[name] = (p[name][0], rs.standard_normal(p[name][1].shape).astype(np.float32) * 0.1)
    for blk in p['blocks']:
        for name in blk:
            blk[name] = (blk[name][0], rs.standard_normal(blk[name][1].shape).astype(np.float32) * 0.1)
    cond = rs.standard_normal((n, cfg['local_dim'] + cfg['global_dim'], T)).astype(np.float32)
    return cfg, p, cond, rs


def test_generation_equals_training_forward_under_teacher_forcing():
    """The queue arithmetic of modules.py:58-74/98-110/232-255 must reproduce the padded, cropped
    convolutions of the training forward (modules.py:40-41, 151-152): feeding the same inputs,
    step i of generate() equals column i of WaveNet.__call__.  The first generated step sees an
    all-zero input vector (generate.py:52), not a one-hot."""
    cfg, p, cond, rs = _gen_setup()
    n, _, T = cond.shape
    forced = rs.randint(0, cfg['input_dim'], (T, n)).astype(np.int32)
    forced[5, 0] = -1                                      # an all-zero input mid-sequence
    u = rs.random_sample((T, n))
    out, logits = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], forced=forced)
    x = np.zeros((n, cfg['input_dim'], T), np.float32)
    for i in range(T - 1):
        for b in range(n):
            if forced[i, b] >= 0:
                x[b, forced[i, b], i + 1] = 1
    y, _ = O.wavenet_fwd(p, x, cond, cfg['n_loop'], cfg['n_layer'])
    np.testing.assert_allclose(logits, y.transpose(2, 0, 1)[:T - 1], rtol=1e-4, atol=1e-5)
    assert out.shape == (n, T) and (out[:, -1] == 0).all()          # generate.py:103-105


def test_choice_from_uniform_is_numpy_choice():
    """Pins the sampler restatement on NumPy itself (the third-party code generate.py:136 calls):
    feeding the double RandomState would draw gives the index RandomState.choice returns."""
    rs = np.random.RandomState(11)
    for trial in range(200):
        logits = (rs.standard_normal((1, 256)) * rs.uniform(0.5, 6)).astype(np.float32)
        pr = O.softmax_axis1(logits)[0]
        seed = int(rs.randint(1 << 30))
        want = np.random.RandomState(seed).choice(256, p=pr)
        u = np.random.RandomState(seed).random_sample()
        assert O.choice_from_uniform(pr, u) == want
    pr = np.array([0.25, 0.25, 0.5], np.float32)
    assert [O.choice_from_uniform(pr, u) for u in (0.0, 0.2499, 0.25, 0.4999, 0.5, 0.999)] == [0, 0, 1, 1, 2, 2]


def test_mol_sampler_known_answers():
    """generate.py:113-133: one dominant component and u = 0.5 give mean/127.5; the result is
    clipped to [-1, 1]; log-scales below log_scale_min are clamped."""
    out = np.zeros((2, 6), np.float32)
    out[:, 0] = 50.0                                        # softmax weight ~1 on component 0
    out[0, 2], out[1, 2] = 63.75, 400.0                     # means
    out[:, 4:] = -100.0                                     # clamped to -40
    v = O.mol_sample_from_uniform(out, np.full((2, 2), 0.5))
    np.testing.assert_allclose(v, [0.5, 1.0], atol=1e-6)
    out[:, 4:] = 2.0
    u = np.array([[0.9, 0.5], [0.1, 0.5]])
    v = O.mol_sample_from_uniform(out, u)
    want0 = (63.75 + np.exp(np.float32(2.0)) * (np.log(0.9) - np.log(0.1))) / 127.5
    np.testing.assert_allclose(v[0], want0, rtol=1e-6)


def test_mol_generation_feeds_back_the_sample():
    cfg, p, cond, rs = _gen_setup(T=10, n=1, input_dim=1, out_dim=6)
    u = rs.uniform(0.05, 0.95, (10, 1, 2))
    out, logits = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], loss_kind='mol')
    assert out.dtype == np.float32 and np.abs(out).max() <= 1 and out[0, -1] == 0
    x = np.zeros((1, 1, 10), np.float32)
    x[0, 0, 1:] = out[0, :-1]
    y, _ = O.wavenet_fwd(p, x, cond, cfg['n_loop'], cfg['n_layer'])
    np.testing.assert_allclose(logits, y.transpose(2, 0, 1)[:9], rtol=1e-4, atol=1e-5)
