import sys, os
sys.path[:0] = ['tests', 'oracle', 'chainer-vq-vae_amd']
import numpy as np, copy
import helpers as H, vqvae_oracle as O
from test_gpu_model import _Iter, _grads_by_name
import test_gpu_configs as TC
import vqvae_amd as V
from vqvae_amd import backend, functions as F
from vqvae_amd.optimizers import Adam
backend.init(0)
which = sys.argv[1]
if which == 'c0':
    cfg = dict(TC.CFG0); T = 7680
    P, model = H.build_model(cfg, seed=0, ema_decay=0.9999)
    P_ema = copy.deepcopy(P['decoder'])
    model.to_gpu(); opt = Adam(2e-4); opt.setup(model)
    batch = O.synth_batch(1, length=T, n_speaker=cfg['n_speaker'], seed=71)
    upd = V.VQVAE_StandardUpdater(_Iter([batch]), opt, device=0)
    sites = H.device_relu_sites(model, batch[0], batch[1], batch[2])
    upd.update()
    with TC._limit_blas():
        losses, cache, G, flips = H.oracle_train_step_aligned(P, {}, batch, cfg['n_loop'], cfg['n_layer'], sites, ema=P_ema, ema_decay=0.9999)
    print('flips', flips)
    g_dev = _grads_by_name(model, opt, True)
    rows = []
    for name, arr in G.items():
        g = g_dev[H._dev_name(name, True)].reshape(arr.shape).astype(np.float64)
        err = np.abs(g - arr); scale = np.abs(arr).max()
        bad = err > 2e-4 * scale
        rows.append((err.max() / scale, TC._rel_l2(g, arr), int(bad.sum()), arr.shape, name,
                     sorted(set(np.argwhere(bad)[:, 0].tolist()))[:6] if bad.any() else []))
    rows.sort(reverse=True)
    for r in rows[:25]: print('%.3e relL2 %.3e nbad %d %s %s rows %s' % r)
else:
    cfg = dict(TC.CFG4); T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    batch = O.synth_batch_raw(1, length=T, n_speaker=cfg['n_speaker'], seed=19)
    def device_step(lazy, bf16=True):
        F.LAZY_CONDITION = lazy
        if bf16: backend.set_matmul_dtype('bfloat16')
        try:
            P, model = H.build_model(cfg, seed=13, use_logistic=True, tweak=TC._mol_conditioned)
            model.to_gpu(); opt = Adam(2e-4); opt.setup(model)
            upd = V.VQVAE_StandardUpdater(_Iter([batch]), opt, device=0); upd.update()
            return P, model.vq._cache[3][0].get(), [float(l.data.get()) for l in upd.last_losses], _grads_by_name(model, opt, False)
        finally:
            F.LAZY_CONDITION = True; backend.set_matmul_dtype('float32')
    P, idx_a, l_a, g_a = device_step(False)
    O.set_bf16(True)
    with TC._limit_blas():
        losses, cache, G = O.train_step(P, {}, batch, cfg['n_loop'], cfg['n_layer'], loss_kind='mol')
    O.set_bf16(False)
    P2 = H.build_model(cfg, seed=13, use_logistic=True, tweak=TC._mol_conditioned)[0]
    with TC._limit_blas():
        losses32, cache32, G32 = O.train_step(P2, {}, batch, cfg['n_loop'], cfg['n_layer'], loss_kind='mol')
    _, idx_b, l_b, g_b = device_step(True)
    _, idx_c, l_c, g_c = device_step(True, bf16=False)
    print('losses oracle bf16', [float(x) for x in losses], 'oracle fp32', [float(x) for x in losses32])
    print('dev full-rate', l_a, 'latent', l_b, 'dev fp32', l_c)
    print('idx flips a/b/c vs bf16 oracle', (idx_a.reshape(-1) != cache['idx'].reshape(-1)).sum(), (idx_b.reshape(-1) != cache['idx'].reshape(-1)).sum(), (idx_c.reshape(-1) != cache32['idx'].reshape(-1)).sum(), 'oracle bf16 vs fp32', (cache['idx'] != cache32['idx']).sum())
    for tag, g_dev, Gref in (('full-rate vs bf16 oracle', g_a, G), ('latent vs bf16 oracle', g_b, G), ('ORACLE bf16 vs ORACLE fp32', None, G32), ('dev fp32 vs oracle fp32', g_c, G32)):
        rows = []
        for name, arr in Gref.items():
            g = (G[name] if g_dev is None else g_dev[H._dev_name(name, False)]).reshape(arr.shape)
            rows.append((TC._rel_l2(g, arr), float(np.abs(g - arr).max() / np.abs(arr).max()), name))
        rows.sort(reverse=True)
        print(tag); 
        for r in rows[:12]: print('   relL2 %.3e maxscaled %.3e %s' % r)
        dec = [r[0] for r in rows if '/decoder' in r[2]]
        print('   decoder tensors: median relL2 %.3e max %.3e' % (np.median(dec), max(dec)))
