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
