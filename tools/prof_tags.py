#!/usr/bin/env python
"""Per-entry-point GPU time of one training step, from the library's HIP-event
profiler (all tags enabled).  Dev tool: python tools/prof_tags.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))
import bench  # noqa: E402
import vqvae_amd as V  # noqa: E402
from vqvae_amd import _lib, backend  # noqa: E402

NAMES = {1: 'resblock gate fwd (K1)', 2: 'resblock res/skip fwd (K2)', 3: 'resblock bwd gz (K3)',
         4: 'resblock bwd gx (K4)', 5: 'resblock bwd gcond (K5)', 6: 'resblock wgrad (K6)',
         7: 'conv fwd', 8: 'conv bwd data', 9: 'conv wgrad', 10: 'vq nearest'}
FLOP = {}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cfg = dict(bench.CFG)
    backend.init(0)
    model, opt = bench.build(cfg, 1)
    model.to_gpu()
    opt.setup(model)
    B = cfg['batch_per_gpu']
    shards = [V.concat_examples(bench.synth_examples(B, cfg, 71), device=0)]
    upd = V.VQVAE_ParallelUpdater(bench.ResidentIterator(shards), opt,
                                  converter=bench.resident_converter, device=0)
    for _ in range(2):
        upd.update()
    backend.synchronize()
    lib = _lib.load()
    lib.vqvae_prof_reset()
    lib.vqvae_prof_enable(-1)
    for _ in range(steps):
        upd.update()
    backend.synchronize()
    lib.vqvae_prof_enable(0)
    BT = B * cfg['length']
    flops = {1: 2.0 * BT * 256 * 512, 2: 2.0 * BT * 512 * 128, 3: 2.0 * BT * 128 * 512,
             4: 2.0 * BT * 256 * 512}
    tot_all = 0
    for tag in sorted(NAMES):
        tot = C.c_double(0)
        cnt = C.c_int(0)
        _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
        if cnt.value == 0:
            continue
        avg = tot.value / cnt.value
        tf = ''
        if tag in flops:
            tf = '  %.1f TFLOP/s' % (flops[tag] / (avg * 1e-3) / 1e12)
        print('%-30s calls/step %6.1f  avg %8.3f ms  per-step %8.3f ms%s'
              % (NAMES[tag], cnt.value / steps, avg, tot.value / steps, tf))
        tot_all += tot.value / steps
    print('sum of tagged: %.2f ms/step' % tot_all)


if __name__ == '__main__':
    main()
