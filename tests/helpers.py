"""Shared helpers for the parity tests: oracle <-> device model plumbing."""
import numpy as np

import vqvae_oracle as O


def to4(a):
    return a.reshape(a.shape + (1,))


def assert_close(got, want, tol=1e-4, name=''):
    """|got - want| <= tol * max(1, |want|)  (north_star: conv/loss outputs within 1e-4 fp32)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    assert got.shape == want.shape, '%s: shape %s vs %s' % (name, got.shape, want.shape)
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    i = int(err.argmax()) if err.size else 0
    assert err.size == 0 or err[i] <= tol, \
        '%s: max err %.3e at %d (got %r want %r) tol %g' % (name, err[i], i, got[i], want[i], tol)


def assert_close_scaled(got, want, tol=1e-4, name=''):
    """Error relative to the tensor's own scale (for gradients whose entries are tiny)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    assert got.shape == want.shape, '%s: shape %s vs %s' % (name, got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30) if want.size else 1.0
    err = np.abs(got - want).max() / scale if want.size else 0.0
    assert err <= tol, '%s: max err %.3e of scale %.3e (tol %g)' % (name, err, scale, tol)


SMALL = dict(d=32, k=64, n_loop=2, n_layer=3, filter_size=2, input_dim=256, residual=64,
             dilated=64, skip=64, out_dim=256, local_dim=32, global_dim=32, n_speaker=7)


MOL = dict(d=32, k=64, n_loop=4, n_layer=7, filter_size=2, input_dim=1, residual=64,
           dilated=64, skip=64, out_dim=30, local_dim=32, global_dim=32, n_speaker=7)   # 28 blocks


def build_model(cfg, seed=0, ema_decay=None, use_logistic=False, tweak=None):
    """Builds (oracle params P, device model) with IDENTICAL initial weights."""
    import vqvae_amd as V
    from vqvae_amd.net import Encoder, ConditionEmbed, VAE
    from vqvae_amd.wavenet import WaveNet
    from vqvae_amd import functions as F
    rs = np.random.RandomState(seed)
    P = O.make_params(rs, **cfg)
    if tweak is not None:
        tweak(P)
    enc = Encoder(cfg['d'])
    wn = WaveNet(cfg['n_loop'], cfg['n_layer'], cfg['filter_size'], cfg['input_dim'],
                 cfg['residual'], cfg['dilated'], cfg['skip'], 256, use_logistic, cfg['out_dim'],
                 -40, cfg['local_dim'] + cfg['global_dim'], 0)
    ce = ConditionEmbed(cfg['n_speaker'], cfg['global_dim'], cfg['local_dim'])
    # force lazy shapes
    for i, ci in zip(range(1, 6), [cfg['d']] + [cfg['local_dim']] * 4):
        getattr(ce, 'local_embed%d' % i)._initialize_params(ci)
    decoder = wn
    if ema_decay is not None:
        decoder = V.ExponentialMovingAverage(wn, ema_decay)
    loss_fun = wn.calculate_logistic_loss if use_logistic else F.softmax_cross_entropy   # train.py:92-95
    model = VAE(enc, decoder, ce, cfg['d'], cfg['k'], 0.25, loss_fun)
    load_params(model, P, ema=ema_decay is not None)
    return P, model


def oracle_named(P):
    """oracle flatten_params names -> arrays, renamed to the device model's namedparams paths."""
    out = {}
    for name, arr in O.flatten_params(P):
        out[name] = arr
    return out


def _dev_name(name, ema):
    # oracle: /decoder/blocks/3/conv/W -> device: /decoder[/target]/resnet/3/conv/W ; /vq/W same
    if name.startswith('/decoder'):
        rest = name[len('/decoder'):]
        rest = rest.replace('/blocks/', '/resnet/')
        return '/decoder' + ('/target' if ema else '') + rest
    return name


def load_params(model, P, ema=False):
    named = dict(model.namedparams())
    for name, arr in O.flatten_params(P):
        dn = _dev_name(name, ema)
        p = named[dn]
        shape = arr.shape + (1,) if arr.ndim == 3 else arr.shape
        p.data = np.ascontiguousarray(arr.reshape(shape), np.float32).copy()
        if ema and dn.startswith('/decoder/target'):
            q = named[dn.replace('/target', '/ema', 1)]
            q.data = p.data.copy()


def device_named_grads(model, ema=False):
    return {n: p for n, p in model.namedparams()}


# --------------------------------------------------------------------------- #
# ReLU kinks.  Two fp32 evaluations of the same network agree to ~1e-7 in every activation, so a
# pre-activation that lands within that distance of zero can come out on either side of the kink.
# One such element out of the ~2M behind each decoder ReLU moves every gradient below it by
# ~1/sqrt(2M) ~ 1e-3 relative (measured at configs[0]: 1e-4..2e-3 of scale), which is the
# subgradient choice at a kink, not arithmetic.  The whole-step parity tests therefore (1) read
# the device's post-ReLU activations, (2) check that the oracle disagrees about "active" ONLY at
# elements that are zero to fp32 noise, and (3) let the oracle's backward use the device's choice
# there.  Everything else -- every multiply-add of forward and backward -- is compared as is.
# --------------------------------------------------------------------------- #
def device_relu_sites(model, x_enc, x_dec, speaker):
    """Post-ReLU activations of the 12 ReLU sites of VAE.__call__ on the device (net.py:20-24,
    49-53; modules.py:155, 158), computed with the model's own links, no graph, no EMA blend."""
    import vqvae_amd as V
    from vqvae_amd import backend, functions as F
    from vqvae_amd.core import Variable
    with V.core.no_backprop_mode():
        h = Variable(backend.to_device(np.ascontiguousarray(x_enc[..., None])))
        enc = []
        for i in range(1, 7):
            h = getattr(model.encoder, 'conv%d' % i)(h, relu=(i < 6))
            if i < 6:
                enc.append(h.data.get()[..., 0])
        e = model.vq(h)
        ce = []
        g = e
        for i in range(1, 6):
            g = getattr(model.condition_embed, 'local_embed%d' % i)(g, relu=True)
            ce.append(g.data.get()[..., 0])
        cond = model.condition_embed(e, Variable(backend.to_device(np.ascontiguousarray(speaker))))
        wn = getattr(model.decoder, 'target', model.decoder)
        xd = Variable(backend.to_device(np.ascontiguousarray(x_dec[..., None])))
        x0 = wn.embed_input(xd)
        s = F.relu(wn.resnet(x0, cond))
        z1 = wn.proj1(s, relu=True)
        return {'enc': enc, 'ce': ce, 's': s.data.get()[..., 0], 'z1': z1.data.get()[..., 0]}


def align_relu_kinks(cache, dev, noise=2e-5, stats=None):
    """Makes the oracle's cached activations take the device's side at ReLU kinks (in place).
    Returns the number of elements changed; fails if the two disagree anywhere that is not zero
    to within ``noise`` x the tensor's scale (that would be an arithmetic difference).

    ``stats`` (a dict) receives the NOISE MODEL's prediction of that number, so that callers can hold the
    count against something derived rather than a round ceiling: per site, sigma = the rms difference of
    the two evaluations over the elements both call active (the arithmetic noise actually present), rho =
    the oracle's density of activations just above zero (elements in (0, 64 sigma] / (64 sigma); the density
    just below is the same to first order), and two evaluations whose difference is ~N(0, sigma^2) disagree
    about the sign of rho * E|delta| = rho * sigma * sqrt(2 / pi) elements on each side of the kink:
    expected = 2 * rho * sigma * sqrt(2 / pi), summed over the sites (stats['expected'], ['per_site'])."""
    flips = [0]
    if stats is not None:
        stats.setdefault('expected', 0.0)
        stats.setdefault('per_site', [])

    def fix(post, dv, name, pre=None):
        ref = post if pre is None else pre
        mism = (ref > 0) != (dv > 0)
        n = int(mism.sum())
        if stats is not None:
            both = (post > 0) & (dv > 0)
            if both.any():
                delta = post[both].astype(np.float64) - dv[both]
                sigma = float(np.sqrt(np.mean(delta * delta)))
                w = 64.0 * sigma
                rho = float(((post > 0) & (post <= w)).sum()) / w if w > 0 else 0.0
                exp = 2.0 * rho * sigma * np.sqrt(2.0 / np.pi)
                stats['expected'] += exp
                stats['per_site'].append((name, n, exp, sigma / max(float(np.abs(ref).max()), 1e-30)))
        if n:
            scale = float(np.abs(ref).max())
            worst = max(float(np.abs(ref[mism]).max()), float(np.abs(dv[mism]).max()))
            assert worst <= noise * scale, \
                '%s: device and oracle disagree about ReLU activity at a value of %.3e (scale %.3e)' % (name, worst, scale)
            on = mism & (dv > 0)
            post[on] = dv[on]
            post[mism & ~on] = 0
            if pre is not None:
                pre[on] = dv[on]
                pre[mism & ~on] = -0.0
            flips[0] += n
    for i in range(5):
        fix(cache['enc_hs'][i + 1], dev['enc'][i], 'encoder relu %d' % (i + 1))
        fix(cache['ce_hs'][i + 1], dev['ce'][i], 'condition_embed relu %d' % (i + 1))
    x, caches, skip_sum, s, z1 = cache['dcache']
    fix(s, dev['s'], 'decoder relu(skip)', pre=skip_sum)
    fix(z1, dev['z1'], 'decoder relu(proj1)')
    return flips[0]


def kink_flip_ceiling(stats):
    """The most ReLU kink flips the noise model allows (see align_relu_kinks): a Poisson count with the predicted mean, held to
    mean + 4 sqrt(mean) + 4 with the mean doubled for what the model leaves out (the noise is neither Gaussian nor
    independent of the value)."""
    m = 2.0 * stats['expected']
    return int(np.ceil(m + 4.0 * np.sqrt(m) + 4.0))


def oracle_train_step_aligned(P, state, batch, n_loop, n_layer, dev_sites, beta=0.25, alpha=2e-4, ema=None,
                              ema_decay=0.9999, loss_kind='softmax', kink_stats=None):
    """oracle.train_step (updaters.py:6-19 incl. the EMA blend) with the ReLU kink choices of the
    device (see above) applied between its forward and its backward."""
    x_enc, x_dec, speaker, t = batch
    losses, cache = O.vae_forward(P, x_enc, x_dec, speaker, t, n_loop, n_layer, beta, loss_kind)
    if ema is not None:
        for (n1, a), (n2, b) in zip(O.flatten_params(ema), O.flatten_params(P['decoder'])):
            O.ema_update(a, b, ema_decay)
    flips = align_relu_kinks(cache, dev_sites, stats=kink_stats)
    G = O.vae_backward(P, cache, speaker, t, n_loop, n_layer, beta, loss_kind)
    flatP, flatG = dict(O.flatten_params(P)), dict(O.flatten_params(G))
    state['t'] = state.get('t', 0) + 1
    for name, p in flatP.items():
        if name not in flatG:
            continue
        m = state.setdefault('m' + name, np.zeros_like(p))
        v = state.setdefault('v' + name, np.zeros_like(p))
        O.adam_update(p, flatG[name], m, v, state['t'], alpha)
    return losses, cache, flatG, flips
