#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  Enqueue time of one training step (no synchronisation
inside) vs its GPU time: the headroom that keeps 8 ranks on one node from becoming host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))
import bench
import vqvae_amd as V
from vqvae_amd import backend
backend.init(0)
cfg = dict(bench.CFG)
model, opt = bench.build(cfg, 1)
model.to_gpu(0); opt.setup(model)
shards = [V.concat_examples(bench.synth_examples(16, cfg, seed=71 + s), device=0) for s in range(2)]
upd = V.VQVAE_ParallelUpdater(bench.ResidentIterator(shards), opt, converter=bench.resident_converter, device=0)
for _ in range(3): upd.update()
backend.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); upd.update(); enq.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
backend.synchronize()
t_all = time.perf_counter() - t0
print('host enqueue per step: median %.2f ms (min %.2f, max %.2f); 10 steps enqueued in %.1f ms, finished on the GPU after %.1f ms' % (
    1e3 * sorted(enq)[5], 1e3 * min(enq), 1e3 * max(enq), 1e3 * t_enq, 1e3 * t_all))
