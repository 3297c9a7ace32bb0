"""CPU oracle: a NumPy restatement of the reference's training hot path (and of
its incremental generation loop).

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``chainer-vq-vae_amd/``)
never imports anything under ``oracle/`` and has no CPU compute path.

Every function restates what the reference (dhgrs/chainer-VQ-VAE, mounted at
/root/reference in the dev container) computes on its Chainer-CPU path and
cites the reference file:line it follows.

Parity pin status
-----------------
* ``vq_forward`` / ``vq_backward`` / ``MuLaw``: PINNED against golden vectors
  produced by executing the reference's own ``utils.py`` bodies
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``choice_from_uniform`` (the sampler of generate.py:136): PINNED against
  ``numpy.random.RandomState.choice`` itself (tests/test_oracle.py); the queue
  arithmetic of incremental generation is checked against the training
  forward (same inputs, step i == column i).
* conv / resize / softmax-CE / MoL / Adam / EMA: the arithmetic lives in
  Chainer 4.0.0b3 (README.md:21), a third-party dependency that is NOT vendored
  under /root/reference and is not installed.  The reference has no tests or
  golden vectors for it.  These functions restate Chainer's published
  algorithms ("[chainer-recalled]" in SURVEY.md) and are checked by analytic
  known-answer tests and fp64 finite-difference gradient checks
  (tests/test_oracle.py) and cross-checked against PyTorch-CPU's implementations of
  the same operators (tests/test_oracle_torch.py): **parity unpinned** for these
  parts -- no vector produced by Chainer itself exists to pin them on.

All functions are dtype-generic (float32 for parity, float64 for gradient
checks).  Internal layout is (B, C, T); the Chainer boundary layout (B, C, T, 1)
is handled by ``squeeze4``/``expand4``.
"""
import numpy as np


# --------------------------------------------------------------------------- #
# layout helpers
# --------------------------------------------------------------------------- #
def squeeze4(x):
    """(B,C,T,1) Chainer NCHW with W=1 (net.py:12; modules.py:13-16) -> (B,C,T)."""
    return x.reshape(x.shape[:3]) if x.ndim == 4 else x


def expand4(x):
    return x.reshape(x.shape + (1,))


# --------------------------------------------------------------------------- #
# mu-law (utils.py:12-29) -- pinned by golden vectors
# --------------------------------------------------------------------------- #
class MuLaw(object):
    """utils.py:12-29."""

    def __init__(self, mu=256, int_type=np.int32, float_type=np.float32):
        self.mu = mu
        self.int_type = int_type
        self.float_type = float_type

    def transform(self, x):
        # utils.py:18-23: compand, then digitize against mu bin edges
        x = x.astype(self.float_type)
        y = np.sign(x) * np.log(1 + self.mu * np.abs(x)) / np.log(1 + self.mu)
        y = np.digitize(y, 2 * np.arange(self.mu) / self.mu - 1) - 1
        return y.astype(self.int_type)

    def itransform(self, y):
        # utils.py:25-29
        y = y.astype(self.float_type)
        y = 2 * y / self.mu - 1
        x = np.sign(y) / self.mu * ((self.mu) ** np.abs(y) - 1)
        return x.astype(self.float_type)


# --------------------------------------------------------------------------- #
# generic 1-D convolution == chainer L.Convolution2D / L.DilatedConvolution2D
# with ksize=(K,1), stride=(s,1), pad=(p,0), dilate=(d,1)
# (call sites net.py:12-17, 34-43; modules.py:13-22, 127-141)  [chainer-recalled]
# --------------------------------------------------------------------------- #
# bf16-operand emulation (BASELINE configs[4]): when enabled, every contraction rounds BOTH
# operands to bfloat16 (round-to-nearest-even) and accumulates in fp32 -- what the HIP path does
# with v_cvt_pk_bf16_f32 + v_mfma_f32_32x32x16_bf16.  Biases and everything else stay fp32.
_BF16 = [False]


# ... and, on the product's default path (ResidualNet's packed chain at the configs' block sizes), keeps the residual
# stream x_l and the gate pre-activation gradients gh_l in HBM as bf16 (vqvae_resblock_desc.storage): `storage=False`
# restates the mode WITHOUT those two roundings (the full-rate-condition path, which does not store them that way)
_STORE16 = [True]


def set_bf16(on, storage=True):
    _BF16[0] = bool(on)
    _STORE16[0] = bool(storage)


def bf16_round(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def _op(a):
    return bf16_round(a) if (_BF16[0] and a.dtype == np.float32) else a


def conv_out_len(L, K, stride, pad, dil):
    return (L + 2 * pad - dil * (K - 1) - 1) // stride + 1


def conv1d_fwd(x, W, b, stride=1, pad=0, dil=1):
    """y[b,o,t] = bias[o] + sum_{c,j} W[o,c,j] * xpad[b,c,t*stride + j*dil - pad].

    x:(B,Ci,L)  W:(Co,Ci,K)  b:(Co,) or None."""
    B, Ci, L = x.shape
    Co, Ci2, K = W.shape
    assert Ci == Ci2
    Lo = conv_out_len(L, K, stride, pad, dil)
    xp = np.pad(_op(x), ((0, 0), (0, 0), (pad, pad)))
    W = _op(W)
    y = np.zeros((B, Co, Lo), dtype=x.dtype)
    for j in range(K):
        xs = xp[:, :, j * dil: j * dil + (Lo - 1) * stride + 1: stride]
        y += np.matmul(np.ascontiguousarray(W[:, :, j]), xs)       # BLAS sgemm per batch item
    if b is not None:
        y += b[None, :, None]
    return y


def conv1d_bwd(x, W, gy, stride=1, pad=0, dil=1, need_gx=True):
    """Returns (gx, gW, gb) of conv1d_fwd."""
    B, Ci, L = x.shape
    Co, _, K = W.shape
    Lo = gy.shape[2]
    xp = np.pad(_op(x), ((0, 0), (0, 0), (pad, pad)))
    gb = gy.sum(axis=(0, 2))
    gy = _op(gy)
    W = _op(W)
    gW = np.zeros_like(W)
    gxp = np.zeros_like(xp) if need_gx else None
    for j in range(K):
        sl = slice(j * dil, j * dil + (Lo - 1) * stride + 1, stride)
        xs = xp[:, :, sl]
        # gW[o,c,j] = sum_{b,t} gy[b,o,t] xs[b,c,t]
        gW[:, :, j] = np.tensordot(gy, xs, axes=((0, 2), (0, 2)))
        if need_gx:
            gxp[:, :, sl] += np.matmul(np.ascontiguousarray(W[:, :, j].T), gy)
    gx = gxp[:, :, pad: pad + L] if need_gx else None
    return gx, gW, gb


def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    # chainer F.sigmoid CPU: tanh(x*0.5)*0.5+0.5  [chainer-recalled]
    half = x.dtype.type(0.5)
    return np.tanh(x * half) * half + half


# --------------------------------------------------------------------------- #
# Encoder (net.py:8-26)
# --------------------------------------------------------------------------- #
def encoder_fwd(p, x):
    """p: dict conv1..conv6 -> (W (Co,Ci,4), b).  x:(B,1,L).  net.py:19-26."""
    hs = [x]
    h = x
    for i in range(1, 7):
        W, b = p['conv%d' % i]
        h = conv1d_fwd(h, W, b, stride=2, pad=1, dil=1)   # net.py:12-17
        if i < 6:
            h = relu(h)                                   # net.py:20-24
        hs.append(h)
    return h, hs


def encoder_bwd(p, hs, gz):
    grads = {}
    g = gz
    for i in range(6, 0, -1):
        W, b = p['conv%d' % i]
        if i < 6:
            g = g * (hs[i] > 0)
        gx, gW, gb = conv1d_bwd(hs[i - 1], W, g, stride=2, pad=1, dil=1,
                                need_gx=(i > 1))
        grads['conv%d' % i] = (gW, gb)
        g = gx
    return grads


# --------------------------------------------------------------------------- #
# VQ / StraightThrough (utils.py:161-255) -- pinned by golden vectors
# --------------------------------------------------------------------------- #
def vq_forward(xs, W):
    """Verbatim arithmetic of StraightThrough.forward, utils.py:176-211.

    xs:(B,d,T',1) or (B,d,T'); W:(k,d).  Returns (embeded (view, same
    non-contiguous transpose as the reference), indexes int32)."""
    e = W
    x = np.expand_dims(xs, 1)                              # utils.py:189
    shape = list(x.shape)
    shape[1] = W.shape[0]
    x = np.broadcast_to(x, tuple(shape))                   # utils.py:190-192
    if x.ndim == 5:
        Wb = np.broadcast_to(np.reshape(W, (1,) + W.shape + (1, 1)), x.shape)
    elif x.ndim == 4:
        Wb = np.broadcast_to(np.reshape(W, (1,) + W.shape + (1,)), x.shape)
    indexes = np.argmin(np.sum((x - Wb) ** 2, axis=2), axis=1).astype(np.int32)  # utils.py:202-203
    embeded = e[indexes]                                   # utils.py:206
    if embeded.ndim == 4:
        embeded = embeded.transpose((0, 3, 1, 2))
    elif embeded.ndim == 3:
        embeded = embeded.transpose((0, 2, 1))
    return embeded, indexes


def vq_forward_chunked(xs, W, chunk=64):
    """Same values as vq_forward, but evaluated per batch-chunk so the
    (B,k,d,T',1) temporaries (utils.py:192-203; 8 GB at the stress shape) fit
    in memory.  Per-row arithmetic and order are unchanged."""
    outs, idxs = [], []
    for s in range(0, xs.shape[0], chunk):
        e, i = vq_forward(xs[s:s + chunk], W)
        outs.append(np.ascontiguousarray(e))
        idxs.append(i)
    return np.concatenate(outs, 0), np.concatenate(idxs, 0)


def vq_backward(indexes, W, gy):
    """StraightThrough.backward, utils.py:213-231.  gx is gy itself
    (utils.py:218-219); gW = onehot^T . gy accumulated in float64 then cast
    (xp.eye default dtype, utils.py:227-228)."""
    g = gy
    if g.ndim == 4:
        g = g.transpose((0, 2, 3, 1))
    elif g.ndim == 3:
        g = g.transpose((0, 2, 1))
    g = g.reshape((-1, g.shape[-1]))
    onehot = np.eye(W.shape[0])[indexes.reshape((-1))]     # float64 (N,k)
    gW = onehot.T.dot(g).astype(gy.dtype)
    return gy, gW


# --------------------------------------------------------------------------- #
# F.resize_images, align-corners bilinear along H with W == 1
# (net.py:54-55, 60-61)  [chainer-recalled, SURVEY Appendix B]
# --------------------------------------------------------------------------- #
def resize_tables(H, outH, dtype=np.float32):
    """Index/weight tables of chainer's resize_images along one axis:
    v = linspace(0, H-1, outH) (float64); v0 = floor(v).clip(0, H-2); v1 = v0+1;
    weights (v1 - v), (v - v0) cast to the array dtype."""
    v = np.linspace(0, H - 1, num=outH)
    v0 = np.floor(v).astype(np.int32)
    v0 = v0.clip(0, H - 2)
    v1 = v0 + 1
    w0 = (v1 - v).astype(dtype)
    w1 = (v - v0).astype(dtype)
    if H == 1:                       # degenerate axis: clip(0,-1) -> -1 (wraps), weight 0/1
        v0 = np.zeros(outH, np.int32)
        v1 = np.zeros(outH, np.int32)
        w0 = np.zeros(outH, dtype)
        w1 = np.ones(outH, dtype)
    return v0, v1, w0, w1


def upsample_fwd(x, outT):
    v0, v1, w0, w1 = resize_tables(x.shape[2], outT, x.dtype)
    return w0[None, None, :] * x[:, :, v0] + w1[None, None, :] * x[:, :, v1]


def upsample_bwd(gy, T):
    v0, v1, w0, w1 = resize_tables(T, gy.shape[2], gy.dtype)
    gx = np.zeros(gy.shape[:2] + (T,), dtype=gy.dtype)
    B, C, _ = gy.shape
    a = (gy * w0[None, None, :]).reshape(B * C, -1)
    b = (gy * w1[None, None, :]).reshape(B * C, -1)
    g2 = gx.reshape(B * C, T)
    for r in range(B * C):
        g2[r] += np.bincount(v0, weights=a[r], minlength=T).astype(gy.dtype)
        g2[r] += np.bincount(v1, weights=b[r], minlength=T).astype(gy.dtype)
    return gx


# --------------------------------------------------------------------------- #
# ConditionEmbed (net.py:29-64)
# --------------------------------------------------------------------------- #
COND_DILS = (1, 2, 4, 8, 16)


def cond_embed_fwd(p, e, speaker, upscale=64):
    """p: local_embed1..5 -> (W (Co,Ci,3), b); 'global_embed' -> (n_speaker, G).
    e:(B,d,T'); speaker:(B,) int32.  Returns cond:(B, Co+G, upscale*T')."""
    hs = [e]
    h = e
    for i, dil in enumerate(COND_DILS):
        W, b = p['local_embed%d' % (i + 1)]
        h = relu(conv1d_fwd(h, W, b, stride=1, pad=dil, dil=dil))   # net.py:34-53
        hs.append(h)
    T = upscale * h.shape[2]
    loc = upsample_fwd(h, T)                                         # net.py:54-55
    g = p['global_embed'][speaker]                                   # net.py:57 EmbedID
    glob = np.broadcast_to(g[:, :, None], g.shape + (T,))            # net.py:58-61
    cond = np.concatenate((loc, glob), axis=1)                       # net.py:63
    return cond, hs


def cond_embed_bwd(p, hs, speaker, gcond, need_ge=True):
    grads = {}
    Cl = hs[-1].shape[1]
    gloc = gcond[:, :Cl]
    gglob = gcond[:, Cl:].sum(axis=2)                # broadcast^T
    gE = np.zeros_like(p['global_embed'])
    np.add.at(gE, speaker, gglob)
    grads['global_embed'] = gE
    g = upsample_bwd(np.ascontiguousarray(gloc), hs[-1].shape[2])
    for i in range(len(COND_DILS), 0, -1):
        dil = COND_DILS[i - 1]
        W, b = p['local_embed%d' % i]
        g = g * (hs[i] > 0)
        gx, gW, gb = conv1d_bwd(hs[i - 1], W, g, stride=1, pad=dil, dil=dil,
                                need_gx=(i > 1 or need_ge))
        grads['local_embed%d' % i] = (gW, gb)
        g = gx
    return grads, g


# --------------------------------------------------------------------------- #
# WaveNet (WaveNet/modules.py)
# --------------------------------------------------------------------------- #
def causal_conv_fwd(x, W, b, dil):
    """DilatedConvolution2D(pad=dil*(K-1)) then crop [:T] (modules.py:13-16, 40-41):
    h[t] = sum_j W[:,:,j] x[t - (K-1-j)*dil] + b, zero for negative times."""
    K = W.shape[2]
    T = x.shape[2]
    return conv1d_fwd(x, W, b, stride=1, pad=dil * (K - 1), dil=dil)[:, :, :T]


def causal_conv_bwd(x, W, gh, dil, need_gx=True):
    K = W.shape[2]
    pad = dil * (K - 1)
    T = x.shape[2]
    g = np.zeros(gh.shape[:2] + (T + pad,), dtype=gh.dtype)   # grad of the crop
    g[:, :, :T] = gh
    return conv1d_bwd(x, W, g, stride=1, pad=pad, dil=dil, need_gx=need_gx)


def gates_saved_as_bf16(Ch, Cr, Cs, T):
    """The bf16 mode of the configs-sized blocks (BASELINE configs[4]) keeps the saved gate values tanh / sigmoid as
    bf16 (csrc/conv_api.hip z_bf16 / GemmArgs::g16): same predicate, restated."""
    return _BF16[0] and Ch == 128 and Cr == 256 and Cs % 256 == 0 and T % 64 == 0


def gh_saved_as_bf16(Ch, Cr, Cs, T, K):
    """... and, on the default (latent-rate condition) path, the gate pre-activation gradient gh = [ga; gb] of every
    block (vqvae_resblock_desc.storage & VQVAE_STORE_GH_BF16; csrc/conv_api.hip bf16_storage_supported): the
    contractions that read it round it anyway, its bias sums and the condition gradient see the rounded values."""
    return gates_saved_as_bf16(Ch, Cr, Cs, T) and K == 2


def resblock_fwd(p, x, cond, dil):
    """ResidualBlock.__call__ (modules.py:30-56), dropout_zero_rate == 0.
    p: conv (W (2*Ch... (Cd,Cr,K), b), condition_proj (W (Cd,Cc,1), b),
       res (W (Cr,Cd/2,1), b), skip (W (Cs,Cd/2,1), b)."""
    Wd, bd = p['conv']
    Wc, bc = p['condition_proj']
    Wr, br = p['res']
    Ws, bs = p['skip']
    h = causal_conv_fwd(x, Wd, bd, dil)                       # modules.py:40-41
    h = h + conv1d_fwd(cond, Wc, bc)                          # modules.py:44
    Ch = h.shape[1] // 2
    ta = np.tanh(h[:, :Ch])                                   # modules.py:47-48
    sb = sigmoid(h[:, Ch:])
    z = ta * sb
    res = conv1d_fwd(z, Wr, br) + x                           # modules.py:52
    if res.dtype == np.float32 and _STORE16[0] and gh_saved_as_bf16(Ch, Wr.shape[0], Ws.shape[0], x.shape[2], Wd.shape[2]) \
            and x.shape[2] % 128 == 0:
        res = bf16_round(res)                                 # the residual stream between the blocks is kept as bf16 (VQVAE_STORE_RES_BF16)
    skip = conv1d_fwd(z, Ws, bs)                              # modules.py:55
    if ta.dtype == np.float32 and gates_saved_as_bf16(Ch, Wr.shape[0], Ws.shape[0], x.shape[2]):
        ta, sb = bf16_round(ta), bf16_round(sb)               # what the backward pass will differentiate
    return res, skip, (x, ta, sb, z)


def resblock_bwd(p, cache, cond, dil, g_res, g_skip, need_gx=True):
    x, ta, sb, z = cache
    Wd, bd = p['conv']
    Wc, bc = p['condition_proj']
    Wr, br = p['res']
    Ws, bs = p['skip']
    grads = {}
    gz = np.zeros_like(z)
    if g_res is not None:
        gzr, gWr, gbr = conv1d_bwd(z, Wr, g_res)
        gz = gz + gzr
        grads['res'] = (gWr, gbr)
    else:
        grads['res'] = None        # last block: residual output unused (modules.py:89-96)
    gzs, gWs, gbs = conv1d_bwd(z, Ws, g_skip)
    gz = gz + gzs
    grads['skip'] = (gWs, gbs)
    one = z.dtype.type(1)
    ga = gz * sb * (one - ta * ta)
    gb_ = gz * ta * sb * (one - sb)
    gh = np.concatenate((ga, gb_), axis=1)
    if gh.dtype == np.float32 and _STORE16[0] and gh_saved_as_bf16(z.shape[1], Wr.shape[0], Ws.shape[0], x.shape[2], Wd.shape[2]):
        gh = bf16_round(gh)
    gc, gWc, gbc = conv1d_bwd(cond, Wc, gh)
    grads['condition_proj'] = (gWc, gbc)
    gx, gWd, gbd = causal_conv_bwd(x, Wd, gh, dil, need_gx=need_gx)
    grads['conv'] = (gWd, gbd)
    if need_gx and g_res is not None:
        gx = gx + g_res
    return gx, gc, grads


def wavenet_dilations(n_loop, n_layer):
    return [2 ** i for i in range(n_layer)] * n_loop          # modules.py:82


def wavenet_fwd(p, x, cond, n_loop, n_layer):
    """WaveNet.__call__ (modules.py:148-160).  p: embed (W (Cr,Cin,2), b),
    blocks [list of resblock dicts], proj1, proj2."""
    We, be = p['embed']
    h = conv1d_fwd(x, We, be, stride=1, pad=1, dil=1)[:, :, :x.shape[2]]   # modules.py:151-152
    caches = []
    skip_sum = None
    for blk, dil in zip(p['blocks'], wavenet_dilations(n_loop, n_layer)):
        h, skip, c = resblock_fwd(blk, h, cond, dil)
        caches.append(c)
        skip_sum = skip if skip_sum is None else skip_sum + skip           # modules.py:92-95
    s = relu(skip_sum)                                                     # modules.py:155
    W1, b1 = p['proj1']
    W2, b2 = p['proj2']
    z1 = relu(conv1d_fwd(s, W1, b1))                                       # modules.py:158
    y = conv1d_fwd(z1, W2, b2)                                             # modules.py:159
    return y, (x, caches, skip_sum, s, z1)


def wavenet_bwd(p, cache, cond, gy, n_loop, n_layer):
    x, caches, skip_sum, s, z1 = cache
    grads = {}
    W1, b1 = p['proj1']
    W2, b2 = p['proj2']
    g, gW2, gb2 = conv1d_bwd(z1, W2, gy)
    grads['proj2'] = (gW2, gb2)
    g = g * (z1 > 0)
    g, gW1, gb1 = conv1d_bwd(s, W1, g)
    grads['proj1'] = (gW1, gb1)
    g_skip = g * (skip_sum > 0)
    dils = wavenet_dilations(n_loop, n_layer)
    g_res = None
    gcond = np.zeros_like(cond)
    bgrads = [None] * len(dils)
    for i in range(len(dils) - 1, -1, -1):
        g_res, gc, bg = resblock_bwd(p['blocks'][i], caches[i], cond, dils[i],
                                     g_res, g_skip, need_gx=True)
        blk = p['blocks'][i]
        if i >= 1 and g_res.dtype == np.float32 and _STORE16[0] and blk['skip'][0].shape[0] == blk['res'][0].shape[0] and \
                g_res.shape[2] % 128 == 0 and gh_saved_as_bf16(blk['conv'][0].shape[0] // 2, blk['res'][0].shape[0],
                                                              blk['skip'][0].shape[0], g_res.shape[2], blk['conv'][0].shape[2]):
            g_res = bf16_round(g_res)      # the residual gradient stream between the blocks is kept as bf16 (VQVAE_STORE_GX_BF16)
        gcond += gc
        bgrads[i] = bg
    grads['blocks'] = bgrads
    We, be = p['embed']
    gpad = np.zeros(g_res.shape[:2] + (g_res.shape[2] + 1,), dtype=g_res.dtype)
    gpad[:, :, :g_res.shape[2]] = g_res
    _, gWe, gbe = conv1d_bwd(x, We, gpad, stride=1, pad=1, dil=1, need_gx=False)
    grads['embed'] = (gWe, gbe)
    return gcond, grads


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def softmax_xent_fwd(y, t):
    """chainer.functions.softmax_cross_entropy(y:(B,q,T), t:(B,T)) with defaults
    (train.py:95, net.py:89): class axis 1, mean over the B*T positions.
    [chainer-recalled: normalize=True, ignore_label=-1 never occurs]"""
    m = y.max(axis=1, keepdims=True)
    ex = np.exp(y - m)
    lse = np.log(ex.sum(axis=1, keepdims=True)) + m
    logp = y - lse
    B, q, T = y.shape
    picked = np.take_along_axis(logp, t[:, None, :].astype(np.int64), axis=1)
    loss = -picked.sum(dtype=y.dtype) / y.dtype.type(B * T)
    return loss, logp


def softmax_xent_bwd(logp, t, gloss=1.0):
    B, q, T = logp.shape
    gy = np.exp(logp)
    np.put_along_axis(gy, t[:, None, :].astype(np.int64),
                      np.take_along_axis(gy, t[:, None, :].astype(np.int64), 1) - 1, axis=1)
    return gy * logp.dtype.type(gloss / (B * T))


def softplus(x):
    # chainer F.softplus(beta=1): max(x,0) + log1p(exp(-|x|))  [chainer-recalled]
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def _mol_terms(y, t, quantize, log_scale_min):
    dt = y.dtype.type
    nr = y.shape[1] // 3
    logit_probs = y[:, :nr]
    means = y[:, nr:2 * nr]
    ls_raw = y[:, 2 * nr:3 * nr]
    log_scales = np.maximum(ls_raw, dt(log_scale_min))                  # modules.py:178-179
    tt = np.broadcast_to(dt(127.5) * t, means.shape)                   # modules.py:181
    centered = tt - means
    inv_std = np.exp(-log_scales)
    half = dt(127.5 / (quantize - 1))
    plus_in = inv_std * (centered + half)                               # modules.py:185
    cdf_plus = sigmoid(plus_in)
    min_in = inv_std * (centered - half)                                # modules.py:187
    cdf_min = sigmoid(min_in)
    log_cdf_plus = plus_in - softplus(plus_in)                          # modules.py:190
    log_one_minus_cdf_min = -softplus(min_in)                           # modules.py:191
    cdf_delta = cdf_plus - cdf_min
    inner = np.log(np.maximum(cdf_delta, dt(1e-12)))                    # modules.py:214-215
    left = tt < dt(127.5 * -0.999)
    right = tt > dt(127.5 * 0.999)
    log_probs = np.where(left, log_cdf_plus, np.where(right, log_one_minus_cdf_min, inner))
    m = logit_probs.max(axis=1, keepdims=True)
    lsm = logit_probs - (np.log(np.exp(logit_probs - m).sum(axis=1, keepdims=True)) + m)
    lp = log_probs + lsm                                                # modules.py:227
    mm = lp.max(axis=1, keepdims=True)
    lse = np.log(np.exp(lp - mm).sum(axis=1, keepdims=True)) + mm
    return dict(nr=nr, ls_raw=ls_raw, inv_std=inv_std, plus_in=plus_in, min_in=min_in,
                cdf_plus=cdf_plus, cdf_min=cdf_min, cdf_delta=cdf_delta, left=left, right=right,
                lsm=lsm, lp=lp, lse=lse)


def mol_loss_fwd(y, t, quantize=256, log_scale_min=-40.0):
    """WaveNet.calculate_logistic_loss (modules.py:169-230).
    y:(B,3*nr_mix,T), t:(B,1,T) float."""
    c = _mol_terms(y, t, quantize, log_scale_min)
    return -c['lse'][:, 0].mean(dtype=y.dtype)                          # modules.py:228-229


def mol_loss_bwd(y, t, quantize=256, log_scale_min=-40.0, gloss=1.0):
    """Analytic gradient of mol_loss_fwd w.r.t. y (what Chainer's autograd yields for
    modules.py:173-229; F.maximum routes the gradient to its first argument where
    x1 >= x2).  Checked against fp64 finite differences in tests/test_oracle.py."""
    dt = y.dtype.type
    c = _mol_terms(y, t, quantize, log_scale_min)
    B, _, T = y.shape
    N = dt(B * T)
    w = np.exp(c['lp'] - c['lse'])                       # responsibilities
    gv = -w / N * dt(gloss)                              # d loss / d (log_probs + lsm)
    g_logit = gv - np.exp(c['lsm']) * gv.sum(axis=1, keepdims=True)
    one = dt(1)
    sp = c['cdf_plus'] * (one - c['cdf_plus'])
    sm = c['cdf_min'] * (one - c['cdf_min'])
    live = c['cdf_delta'] >= dt(1e-12)
    inv_d = np.where(live, one / np.maximum(c['cdf_delta'], dt(1e-12)), dt(0))
    d_plus = np.where(c['left'], one - c['cdf_plus'], np.where(c['right'], dt(0), inv_d * sp))
    d_min = np.where(c['left'], dt(0), np.where(c['right'], -c['cdf_min'], -inv_d * sm))
    g_mean = gv * (d_plus + d_min) * (-c['inv_std'])
    g_ls = gv * (d_plus * (-c['plus_in']) + d_min * (-c['min_in']))
    g_ls = np.where(c['ls_raw'] >= dt(log_scale_min), g_ls, dt(0))
    return np.concatenate((g_logit, g_mean, g_ls), axis=1)


# --------------------------------------------------------------------------- #
# incremental generation (WaveNet/modules.py:58-74, 98-110, 232-255;
# generate.py:105-145) -- SURVEY section 8f row 2
# --------------------------------------------------------------------------- #
def wavenet_initialize(p, n, n_loop, n_layer, dtype=np.float32):
    """WaveNet.initialize(n) (modules.py:232-244) -> ResidualNet.initialize (98-100) ->
    ResidualBlock.initialize (58-67): all-zero queues; the convs lose their padding.
    (proj1_queue / proj2_queue3 have length 1: the 1x1 convs see the current value only.)"""
    We = p['embed'][0]
    R = We.shape[0]
    st = {'embed_queue': np.zeros((n, We.shape[1], 2), dtype), 'queues': []}
    for dil in wavenet_dilations(n_loop, n_layer):
        K = p['blocks'][0]['conv'][0].shape[2]
        st['queues'].append(np.zeros((n, R, dil * (K - 1) + 1), dtype))
    return st


def wavenet_generate_step(p, st, x, cond_t, n_loop, n_layer):
    """WaveNet.generate(x, condition) (modules.py:246-255): x (n, input_dim, 1),
    cond_t (n, cond_dim, 1) -> y (n, out_dim, 1).  Queues are updated in place of ``st``."""
    We, be = p['embed']
    st['embed_queue'] = np.concatenate((st['embed_queue'][:, :, 1:], x), axis=2)     # modules.py:247
    h = conv1d_fwd(st['embed_queue'], We, be, stride=1, pad=0, dil=1)                # pad (0,0): 234
    skip_sum = None
    for i, (blk, dil) in enumerate(zip(p['blocks'], wavenet_dilations(n_loop, n_layer))):
        # push (modules.py:71-74); the condition queue has length 1 => it IS cond_t
        st['queues'][i] = np.concatenate((st['queues'][i][:, :, 1:], h), axis=2)
        q = st['queues'][i]
        # pop (modules.py:68-69) = __call__(queue, condition_queue) with conv.pad = 0 (64)
        Wd, bd = blk['conv']
        Wc, bc = blk['condition_proj']
        Wr, br = blk['res']
        Ws, bs = blk['skip']
        hh = conv1d_fwd(q, Wd, bd, stride=1, pad=0, dil=dil)          # length dil+1 -> 1 (40-41)
        hh = hh + conv1d_fwd(cond_t, Wc, bc)                          # modules.py:44
        Ch = hh.shape[1] // 2
        z = np.tanh(hh[:, :Ch]) * sigmoid(hh[:, Ch:])                 # modules.py:47-48
        h = conv1d_fwd(z, Wr, br) + q[:, :, -1:]                      # modules.py:54 (lengths differ)
        skip = conv1d_fwd(z, Ws, bs)                                  # modules.py:55
        skip_sum = skip if skip_sum is None else skip_sum + skip      # modules.py:105-109
    s = relu(skip_sum)                                                # modules.py:249
    W1, b1 = p['proj1']
    W2, b2 = p['proj2']
    z1 = relu(conv1d_fwd(s, W1, b1))                                  # modules.py:251-252
    return conv1d_fwd(z1, W2, b2)                                     # modules.py:254-255


def softmax_axis1(y):
    """chainer.functions.softmax (axis=1): exp(y - max) / sum, in the input dtype. [chainer-recalled]"""
    e = np.exp(y - y.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def choice_from_uniform(p, u):
    """numpy.random.choice(len(p), p=p) (generate.py:136-138) for the uniform double ``u`` it
    draws: NumPy's legacy RandomState.choice converts p to float64, builds cdf = cumsum(p),
    normalises by cdf[-1] and returns cdf.searchsorted(random_sample(), side='right').
    Pinned in tests/test_oracle.py against numpy.random.RandomState itself."""
    cdf = np.asarray(p, np.float64).cumsum()
    cdf /= cdf[-1]
    return int(cdf.searchsorted(u, side='right'))


def mol_sample_from_uniform(out, u, log_scale_min=-40.0):
    """generate.py:113-133 for one step: out (n, 3*nr_mix) float32, u (n, nr_mix) float64 in (0,1)
    (what xp.random.uniform(0, 1, shape) returned).  NOTE the reference does not pick a mixture
    component: it averages one logistic sample per component with the softmax weights."""
    nr = out.shape[1] // 3
    logit_probs = out[:, :nr]
    means = out[:, nr:2 * nr]
    log_scales = np.maximum(out[:, 2 * nr:3 * nr], out.dtype.type(log_scale_min))
    scales = np.exp(log_scales)
    rand = means + scales * (np.log(u) - np.log(1 - u))               # float64
    rand = rand * softmax_axis1(logit_probs)
    value = rand.sum(axis=1).astype(np.float32)
    value /= 127.5
    return np.clip(value, -1, 1)


def wavenet_generate(p, cond, uniforms, n_loop, n_layer, loss_kind='softmax', quantize=256,
                     log_scale_min=-40.0, forced=None, n_steps=None):
    """The sampling loop of generate.py:101-145 (n sequences in lockstep; the reference runs
    n = 1).  cond (n, cond_dim, T); uniforms (T, n) [softmax] or (T, n, nr_mix) [mol] are the
    doubles the reference would draw from numpy's global RNG.  ``forced`` (T, n) replaces the
    fed-back sample (teacher forcing; -1 = all-zero input for the softmax kind).
    Returns (output (n, T), logits (T, n, out_dim)); output[:, T-1] stays 0 (generate.py:103)."""
    n, _, T = cond.shape
    steps = T - 1 if n_steps is None else n_steps
    input_dim = p['embed'][0].shape[1]
    st = wavenet_initialize(p, n, n_loop, n_layer, cond.dtype)
    x = np.zeros((n, input_dim, 1), cond.dtype)                       # generate.py:52
    out_dim = p['proj2'][0].shape[0]
    output = np.zeros((n, T), np.int32 if loss_kind == 'softmax' else np.float32)
    logits = np.zeros((steps, n, out_dim), cond.dtype)
    for i in range(steps):
        y = wavenet_generate_step(p, st, x, cond[:, :, i:i + 1], n_loop, n_layer)[:, :, 0]
        logits[i] = y
        if loss_kind == 'softmax':
            pr = softmax_axis1(y)
            value = np.array([choice_from_uniform(pr[b], uniforms[i, b]) for b in range(n)])
            output[:, i] = value
            nxt = value if forced is None else forced[i]
            x = np.zeros((n, input_dim, 1), cond.dtype)               # generate.py:139-141
            for b in range(n):
                if nxt[b] >= 0:
                    x[b, nxt[b], 0] = 1
        else:
            value = mol_sample_from_uniform(y, uniforms[i], log_scale_min)
            output[:, i] = value
            nxt = value if forced is None else forced[i]
            x = np.asarray(nxt, cond.dtype).reshape(n, 1, 1) * np.ones((1, input_dim, 1), cond.dtype)
    return output, logits


# --------------------------------------------------------------------------- #
# optimizer / EMA
# --------------------------------------------------------------------------- #
def adam_update(param, grad, m, v, t, alpha, beta1=0.9, beta2=0.999, eps=1e-8):
    """chainer.optimizers.Adam update rule (train.py:101-102) [chainer-recalled]:
    m += (1-b1)(g-m); v += (1-b2)(g*g-v);
    p -= alpha*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).  In place; t is the
    1-based step count."""
    dt = param.dtype.type
    m += dt(1 - beta1) * (grad - m)
    v += dt(1 - beta2) * (grad * grad - v)
    fix1 = 1.0 - beta1 ** t
    fix2 = 1.0 - beta2 ** t
    lr = dt(alpha * np.sqrt(fix2) / fix1)
    param -= lr * m / (np.sqrt(v) + dt(eps))


def ema_update(ema, target, decay):
    """ExponentialMovingAverage (utils.py:151-155): ema <- decay*target + (1-decay)*ema."""
    dt = ema.dtype.type
    ema[...] = dt(decay) * target + dt(1 - decay) * ema


# --------------------------------------------------------------------------- #
# parameter construction  (Chainer link defaults: LeCunNormal W, zero b,
# EmbedID N(0,1); utils.py:244)  [chainer-recalled]
# --------------------------------------------------------------------------- #
def _lecun(rs, shape, dtype):
    fan_in = int(np.prod(shape[1:]))
    return (rs.standard_normal(shape) / np.sqrt(fan_in)).astype(dtype)


def make_params(rs, d=64, k=512, n_loop=2, n_layer=10, filter_size=2, input_dim=256,
                residual=256, dilated=256, skip=256, out_dim=256,
                local_dim=64, global_dim=128, n_speaker=109, dtype=np.float32):
    """Builds the full parameter tree in Chainer shapes (trailing W=1 axis dropped)."""
    def conv(co, ci, K):
        return (_lecun(rs, (co, ci, K), dtype), np.zeros(co, dtype))
    P = {}
    enc = {}
    ci = 1
    for i in range(1, 7):
        enc['conv%d' % i] = conv(d, ci, 4)
        ci = d
    P['encoder'] = enc
    P['vq'] = _lecun(rs, (k, d), dtype)
    ce = {}
    ci = d
    for i in range(1, 6):
        ce['local_embed%d' % i] = conv(local_dim, ci, 3)
        ci = local_dim
    ce['global_embed'] = rs.standard_normal((n_speaker, global_dim)).astype(dtype)
    P['condition_embed'] = ce
    cdim = local_dim + global_dim
    dec = {'embed': conv(residual, input_dim, 2), 'blocks': []}
    for _ in wavenet_dilations(n_loop, n_layer):
        dec['blocks'].append({
            'conv': conv(dilated, residual, filter_size),
            'condition_proj': conv(dilated, cdim, 1),
            'res': conv(residual, dilated // 2, 1),
            'skip': conv(skip, dilated // 2, 1)})
    dec['proj1'] = conv(skip, skip, 1)
    dec['proj2'] = conv(out_dim, skip, 1)
    P['decoder'] = dec
    return P


def flatten_params(P, prefix=''):
    """Deterministic (name, array) list; names follow chainer's namedparams()
    paths (generate.py:67-81 key layout)."""
    out = []
    if isinstance(P, dict):
        for key in P:
            out += flatten_params(P[key], prefix + '/' + key)
    elif isinstance(P, list):
        for i, v in enumerate(P):
            out += flatten_params(v, prefix + '/%d' % i)
    elif isinstance(P, tuple):
        out.append((prefix + '/W', P[0]))
        out.append((prefix + '/b', P[1]))
    elif P is None:
        pass
    else:
        out.append((prefix + '/W', P))
    return out


# --------------------------------------------------------------------------- #
# VAE forward + the three-loss backward of the updaters
# --------------------------------------------------------------------------- #
def vae_forward(P, x_enc, x_dec, speaker, t, n_loop, n_layer, beta=0.25, loss_kind='softmax',
                quantize=256, log_scale_min=-40.0):
    """VAE.__call__ (net.py:79-96).  loss_kind 'softmax': softmax-CE (train.py:95),
    x_dec:(B,q,T) one-hot, t:(B,T) int32.  loss_kind 'mol': discretised mixture of
    logistics (train.py:93, modules.py:169-230), x_dec:(B,1,T) raw, t:(B,1,T) float."""
    z, enc_hs = encoder_fwd(P['encoder'], x_enc)                       # net.py:81
    e4, idx = vq_forward(expand4(z), P['vq'])                          # net.py:82 (and 83: same values)
    e = np.ascontiguousarray(squeeze4(e4))
    cond, ce_hs = cond_embed_fwd(P['condition_embed'], e, speaker)     # net.py:85
    y, dcache = wavenet_fwd(P['decoder'], x_dec, cond, n_loop, n_layer)  # net.py:86
    if loss_kind == 'softmax':
        loss1, logp = softmax_xent_fwd(y, t)                           # net.py:89
    else:
        loss1, logp = mol_loss_fwd(y, t, quantize, log_scale_min), None
    diff = z - e
    loss2 = np.mean(diff ** 2, dtype=z.dtype)                          # net.py:90
    loss3 = z.dtype.type(beta) * np.mean(diff ** 2, dtype=z.dtype)     # net.py:91
    cache = dict(z=z, enc_hs=enc_hs, idx=idx, e=e, cond=cond, ce_hs=ce_hs,
                 dcache=dcache, logp=logp, y=y)
    return (loss1, loss2, loss3), cache


def vae_backward(P, cache, speaker, t, n_loop, n_layer, beta=0.25, loss_kind='softmax',
                 quantize=256, log_scale_min=-40.0):
    """Gradient of the updater's sequence (updaters.py:13-19):
    cleargrads; loss1.backward(); vq.cleargrads(); loss2.backward(); loss3.backward().
    Net effect: decoder, condition_embed <- dloss1; encoder <- dloss1 (straight
    through, utils.py:218-219) + dloss3; vq.W <- dloss2 only."""
    z, e = cache['z'], cache['e']
    G = {}
    if loss_kind == 'softmax':
        gy = softmax_xent_bwd(cache['logp'], t)
    else:
        gy = mol_loss_bwd(cache['y'], t, quantize, log_scale_min)
    gcond, G['decoder'] = wavenet_bwd(P['decoder'], cache['dcache'], cache['cond'],
                                      gy, n_loop, n_layer)
    G['condition_embed'], ge = cond_embed_bwd(P['condition_embed'], cache['ce_hs'],
                                              speaker, gcond, need_ge=True)
    # loss1 -> straight-through -> encoder (gx = gy, utils.py:218-219)
    gz = ge
    # loss3 = beta*mean((z - sg(e))^2) -> encoder
    n = z.size
    gz = gz + z.dtype.type(beta) * z.dtype.type(2.0 / n) * (z - e)
    G['encoder'] = encoder_bwd(P['encoder'], cache['enc_hs'], gz)
    # loss2 = mean((sg(z) - e_)^2) -> W only
    ge_ = z.dtype.type(-2.0 / n) * (z - e)
    _, gW = vq_backward(cache['idx'], P['vq'], expand4(ge_))
    G['vq'] = gW
    return G


def train_step(P, state, batch, n_loop, n_layer, beta=0.25, alpha=2e-4, ema=None,
               ema_decay=0.9999, grad_sum_hook=None, loss_kind='softmax'):
    """One VQVAE_StandardUpdater.update_core (updaters.py:6-19) incl. the EMA
    blend that runs at forward time (utils.py:146-155).  ``state`` holds Adam
    m, v per flattened param name and the step count.  ``grad_sum_hook`` lets a
    test insert the data-parallel sum (updaters.py:71-72)."""
    x_enc, x_dec, speaker, t = batch
    losses, cache = vae_forward(P, x_enc, x_dec, speaker, t, n_loop, n_layer, beta, loss_kind)
    if ema is not None:                      # EMA.__call__ runs right after target fwd
        for (n1, a), (n2, b) in zip(flatten_params(ema), flatten_params(P['decoder'])):
            ema_update(a, b, ema_decay)
    G = vae_backward(P, cache, speaker, t, n_loop, n_layer, beta, loss_kind)
    flatP = dict(flatten_params(P))
    flatG = dict(flatten_params(G))
    if grad_sum_hook is not None:
        flatG = grad_sum_hook(flatG)
    state['t'] = state.get('t', 0) + 1
    for name, p in flatP.items():
        if name not in flatG:               # e.g. last block's res conv: grad is None -> skipped
            continue
        m = state.setdefault('m' + name, np.zeros_like(p))
        v = state.setdefault('v' + name, np.zeros_like(p))
        adam_update(p, flatG[name], m, v, state['t'], alpha)
    return losses, cache, flatG


# --------------------------------------------------------------------------- #
# synthetic inputs reproducing Preprocess's output contract (utils.py:54-110)
# --------------------------------------------------------------------------- #
def synth_batch(B, length=7680, quantize=256, n_speaker=109, seed=71, sr=16000,
                dtype=np.float32):
    """sum of 3 sinusoids + noise, peak-normalised (utils.py:58),
    mu-law (utils.py:62), one-hot[:, :-1] (utils.py:85-87,102), t = q[1:]
    (utils.py:109).  Returns (x_enc (B,1,L+1), x_dec (B,q,L), speaker (B,), t (B,L))."""
    rs = np.random.RandomState(seed)
    L = length + 1
    n = np.arange(L) / float(sr)
    raws, qs = [], []
    mu = MuLaw(quantize)
    for _ in range(B):
        f = rs.uniform(80, 4000, 3)
        ph = rs.uniform(0, 2 * np.pi, 3)
        a = rs.uniform(0.2, 1.0, 3)
        raw = sum(a[i] * np.sin(2 * np.pi * f[i] * n + ph[i]) for i in range(3))
        raw = raw + 0.05 * rs.standard_normal(L)
        raw = (raw / np.abs(raw).max()).astype(np.float32)
        raws.append(raw)
        qs.append(mu.transform(raw))
    raw = np.stack(raws)[:, None, :].astype(dtype)
    q = np.stack(qs)
    x_dec = np.identity(quantize, dtype=dtype)[q[:, :-1]].transpose(0, 2, 1)
    t = q[:, 1:].astype(np.int32)
    speaker = rs.randint(0, n_speaker, size=B).astype(np.int32)
    return raw, np.ascontiguousarray(x_dec), speaker, t


def preprocess_contract(raw, length, quantize=256, start=0, mu_law_input=True, use_logistic=False,
                        speaker_id=0):
    """The array half of Preprocess.__call__ (utils.py:57-110) for an already loaded and trimmed
    waveform: peak-normalise, mu-law, pad with zeros / bin quantize//2 or crop `length+1`
    samples at `start`, one-hot, and the 4-tuple (raw, x_dec, speaker, t)."""
    L = length + 1
    raw = raw / np.abs(raw).max()                                         # utils.py:58
    raw = raw.astype(np.float32)
    q = MuLaw(quantize).transform(raw) if (mu_law_input or not use_logistic) else None   # utils.py:62
    if len(raw) <= L:                                                     # utils.py:66-75
        pad = L - len(raw)
        raw = np.concatenate((raw, np.zeros(pad, dtype=np.float32)))
        if q is not None:
            q = np.concatenate((q, quantize // 2 * np.ones(pad))).astype(np.int32)
    else:                                                                 # utils.py:76-81
        raw = raw[start:start + L]
        if q is not None:
            q = q[start:start + L]
    raw4 = raw[None, :, None]                                             # utils.py:88-89
    if mu_law_input:
        one_hot = np.identity(quantize, dtype=np.float32)[q]              # utils.py:85-87
        x_dec = np.expand_dims(one_hot.T, 2)[:, :-1]
    else:
        x_dec = raw4[:, :-1]
    t = raw4[:, 1:] if use_logistic else np.expand_dims(q, 1)[1:]         # utils.py:105-109
    return raw4, x_dec, np.array(speaker_id, dtype=np.int32), t


def synth_batch_raw(B, length=7680, n_speaker=109, seed=71, dtype=np.float32):
    """The use_logistic / input_dim=1 variant of Preprocess's contract
    (utils.py:104, 107): x_dec = raw[:, :-1], t = raw[:, 1:], both (B,1,L) float."""
    raw, _, speaker, _ = synth_batch(B, length, 256, n_speaker, seed, dtype=dtype)
    return raw, np.ascontiguousarray(raw[:, :, :-1]), speaker, np.ascontiguousarray(raw[:, :, 1:])
