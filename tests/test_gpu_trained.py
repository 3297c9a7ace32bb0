"""Parity at a TRAINED state (updaters.py:13-19, WaveNet/modules.py:40-55).

Every other whole-step test starts from LeCun-normal weights.  `float32x2` splits the residual stream x_l and the gate
gradients gh_l under A-PRIORI bounds built from weight norms: their looseness -- every power of two
costs a bit of the mode's absolute floor -- is only known where somebody looked, and saturated gates / peaky soft-max
gradients only exist after training.  So: the configs[0]-sized model (20 blocks, 256 channels, d = 64, k = 512) is
trained ON THE DEVICE for 300 Adam steps (fp32 MFMA mode, four synthetic minibatches, fixed seeds, twice train.py's step
size).  On this synthetic data (three sinusoids + noise per clip) the quantiser's assignment collapses onto one code while
the decoder learns -- posterior collapse, a property of the model, printed by the test; the VQ path's own hard cases (ties,
near-ties, the 8 192-code stress) are pinned by the reference-generated goldens.  What the trained state does bring:
saturated gates (up to 44 % of a block's tanh values within 1e-3 of +-1), a peaky soft-max, weight norms 1.5x the initial ones, its parameters are read
back, and ONE whole step from that snapshot runs in each fp32 matmul mode against the oracle loaded with the same
snapshot, at the configs' bars: argmin indices bit-exact, three losses 1e-4, every gradient 1e-4 of its scale,
every parameter after Adam 1e-4, EMA 1e-5.  In `float32x2` the test also reads the chain's own bookkeeping -- the
bound each pre-split tensor was split under and the maximum its producer published -- prints log2(bound / max) per
block and holds it to the run-time guard's limit (backend.CONTRACT_LOG2_LIMIT), and checks that the guard counted no
violation."""
import copy

import numpy as np
import pytest

import helpers as H
import vqvae_oracle as O
from helpers import assert_close
from test_gpu_configs import CFG0, _limit_blas
from test_gpu_model import _Iter, _grads_by_name

pytestmark = pytest.mark.gpu

T = 7680
TRAIN_STEPS = 300
TRAIN_LR = 4e-4


@pytest.fixture(scope='module')
def trained(gpu):
    """(P, P_ema, how far training moved things): host copies of the parameters after TRAIN_STEPS device steps."""
    import vqvae_amd as V
    from vqvae_amd.optimizers import Adam
    cfg = dict(CFG0)
    gpu.set_matmul_dtype('float32')
    try:
        P, model = H.build_model(cfg, seed=5, ema_decay=0.9999)
        P0 = copy.deepcopy(P)
        model.to_gpu()
        opt = Adam(TRAIN_LR)
        opt.setup(model)
        batches = [O.synth_batch(1, length=T, n_speaker=cfg['n_speaker'], seed=300 + s) for s in range(4)]
        upd = V.VQVAE_StandardUpdater(_Iter(batches), opt, device=0)
        first = last = None
        for step in range(TRAIN_STEPS):
            upd.update()
            if step in (0, TRAIN_STEPS - 1):
                l = [float(v.data.get()) for v in upd.last_losses]
                first, last = (l, last) if step == 0 else (first, l)
        named = dict(model.namedparams())
        for name, arr in O.flatten_params(P):
            arr[...] = named[H._dev_name(name, True)].data.get().reshape(arr.shape)
        P_ema = copy.deepcopy(P['decoder'])
        for name, arr in O.flatten_params(P_ema):
            arr[...] = named['/decoder/ema' + name.replace('/blocks/', '/resnet/')].data.get().reshape(arr.shape)
    finally:
        gpu.set_matmul_dtype(gpu.default_matmul_dtype())
    moved = {}
    for (name, a), (_, b) in zip(O.flatten_params(P), O.flatten_params(P0)):
        if name.endswith('/W') and np.abs(b).max() > 0:          # (biases start at zero)
            moved[name] = float(np.abs(a - b).max() / np.abs(b).max())
    assert np.isfinite(last).all() and last[0] < first[0] - 0.5, (first, last)      # it did train
    print('trained state: %d steps at lr %g, loss1 %.3f -> %.3f, loss2 %.4f -> %.4f; largest move of a weight tensor / its initial scale: %.2f (median %.2f)'
          % (TRAIN_STEPS, TRAIN_LR, first[0], last[0], first[1], last[1], max(moved.values()), float(np.median(list(moved.values())))))
    return P, P_ema, (first, last)


def _looseness(nb):
    """log2(bound / actual maximum) of every pre-split tensor of the last backward sweep, from the chain's own words
    (wavenet.ResidualStackFunction._slot: x_l at l, gh_l at nb + l, scale words of x_l at 3 nb + 1 + l, of gh_l at 4 nb + 1 + l)."""
    from vqvae_amd import wavenet
    n, words = wavenet.LAST_CONTRACT_WORDS
    assert n == nb

    def f(u):
        return float(np.array([u], np.uint32).view(np.float32)[0])
    lx, lg = {}, {}
    for l in range(nb):
        bx, mx = f(words[3 * nb + 1 + l][0]), f(words[l].max())
        if bx > 0 and mx > 0:
            lx[l] = float(np.log2(bx / mx))
        bg, mg = f(words[4 * nb + 1 + l][0]), f(words[nb + l].max())
        if bg > 0 and mg > 0:
            lg[l] = float(np.log2(bg / mg))
    return lx, lg


def test_one_step_from_a_trained_snapshot_matches_oracle(gpu, matmul_mode, trained):
    import vqvae_amd as V
    from vqvae_amd import backend, wavenet
    from vqvae_amd.optimizers import Adam
    cfg = dict(CFG0)
    nb = cfg['n_loop'] * cfg['n_layer']
    P_tr, P_ema_tr, _ = trained
    src = dict(O.flatten_params(P_tr))

    def load(Pn):
        for name, arr in O.flatten_params(Pn):
            arr[...] = src[name]
    P, model = H.build_model(cfg, seed=5, ema_decay=0.9999, tweak=load)
    P_ema = copy.deepcopy(P_ema_tr)
    named = dict(model.namedparams())
    for name, arr in O.flatten_params(P_ema):           # the EMA copy is its own trained state, not a copy of the target
        q = named['/decoder/ema' + name.replace('/blocks/', '/resnet/')]
        q.data = np.ascontiguousarray(arr.reshape(q.data.shape), np.float32).copy()
    model.to_gpu()
    opt = Adam(2e-4)
    opt.setup(model)
    batch = O.synth_batch(1, length=T, n_speaker=cfg['n_speaker'], seed=301)          # one of the training minibatches
    upd = V.VQVAE_StandardUpdater(_Iter([batch]), opt, device=0)
    sites = H.device_relu_sites(model, batch[0], batch[1], batch[2])
    backend.f32x2_contract_violations(reset=True)
    wavenet.KEEP_CONTRACT_WORDS = True
    wavenet.LAST_CONTRACT_WORDS = None
    try:
        upd.update()
    finally:
        wavenet.KEEP_CONTRACT_WORDS = False
    ks = {}
    with _limit_blas():
        losses, cache, G, flips = H.oracle_train_step_aligned(P, {}, batch, cfg['n_loop'], cfg['n_layer'], sites,
                                                              ema=P_ema, ema_decay=0.9999, kink_stats=ks)
    print('trained (%s): %d ReLU kink elements took the other side on the device; the noise model predicts %.1f, allows %d'
          % (matmul_mode, flips, ks['expected'], H.kink_flip_ceiling(ks)))
    assert flips <= H.kink_flip_ceiling(ks), ks['per_site']
    idx_dev = model.vq._cache[3][0].get()
    np.testing.assert_array_equal(idx_dev.reshape(cache['idx'].shape), cache['idx'])
    print('trained (%s): %d distinct codes among the %d latents' % (matmul_mode, len(np.unique(cache['idx'])), cache['idx'].size))
    l_dev = [float(l.data.get()) for l in upd.last_losses]
    for i, (a, b) in enumerate(zip(l_dev, losses)):
        assert_close(a, float(b), 1e-4, 'trained loss%d' % (i + 1))
    g_dev = _grads_by_name(model, opt, True)
    over = []
    for name, arr in G.items():
        dn = H._dev_name(name, True)
        got = g_dev[dn].reshape(arr.shape).astype(np.float64)
        err = float(np.abs(got - arr).max() / max(np.abs(arr).max(), 1e-30))
        over.append((err, dn))
        assert err <= 1e-4, 'trained (%s) grad %s: %.3e of scale (north_star: 1e-4)' % (matmul_mode, dn, err)
    print('trained (%s): losses %s; worst of %d gradient tensors, of scale: %s'
          % (matmul_mode, ['%.5f' % v for v in l_dev], len(G), sorted(over, reverse=True)[:3]))
    named = dict(model.namedparams())
    for name, arr in O.flatten_params(P):
        dn = H._dev_name(name, True)
        assert_close(named[dn].data.get().reshape(arr.shape), arr, 1e-4, 'trained param ' + dn)
    for name, arr in O.flatten_params(P_ema):
        dn = '/decoder/ema' + name.replace('/blocks/', '/resnet/')
        assert_close(named[dn].data.get().reshape(arr.shape), arr, 1e-5, 'trained ema ' + dn)
    # how saturated the trained gates are (what initialisation never shows): the share of sigmoid values within 1e-3 of 0 / 1
    sat = []
    tsat = []
    for c in cache['dcache'][1]:                        # per block: (x, tanh, sigmoid, z), oracle.resblock_fwd
        ta, sg = c[1], c[2]
        sat.append(float(((sg < 1e-3) | (sg > 1 - 1e-3)).mean()))
        tsat.append(float((np.abs(ta) > 1 - 1e-3).mean()))
    print('trained: share of saturated gate values per block (within 1e-3 of the asymptote): sigmoid max %.4f mean %.4f, tanh max %.4f mean %.4f'
          % (max(sat), float(np.mean(sat)), max(tsat), float(np.mean(tsat))))
    if matmul_mode == 'float32x2':
        lx, lg = _looseness(nb)
        assert len(lx) == nb - 1 and len(lg) == nb, (sorted(lx), sorted(lg))
        print('trained: log2(bound / max) of the pre-split residual stream x_l, l = 1..%d: %s' % (nb - 1, ' '.join('%.1f' % lx[l] for l in sorted(lx))))
        print('trained: log2(bound / max) of the pre-split gate gradients gh_l, l = 0..%d: %s' % (nb - 1, ' '.join('%.1f' % lg[l] for l in sorted(lg))))
        lim = backend.CONTRACT_LOG2_LIMIT
        assert all(-0.002 <= v <= lim for v in lx.values()) and all(-0.002 <= v <= lim for v in lg.values()), (lx, lg)
        rep = backend.f32x2_contract_violations()
        assert rep['violations'] == 0 and rep['checked'] == 2 * nb - 1, rep
        assert abs(rep['worst_log2'] - max(list(lx.values()) + list(lg.values()))) < 1e-3, rep
    else:
        assert wavenet.LAST_CONTRACT_WORDS is None
