"""Per-call-site GPU time of one training step at configs[1] (the library's own HIP-event profiler,
csrc/common.h ProfScope): which entry points carry the step.  usage: python tools/tag_times.py [mode]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'chainer-vq-vae_amd')]
import bench
import vqvae_amd as V
from vqvae_amd import _lib, backend
from vqvae_amd.comm import SingleCommunicator

backend.init(0)
if len(sys.argv) > 1:
    backend.set_matmul_dtype(sys.argv[1])
backend.set_overlap(False)
cfg = dict(bench.CFG)
model, opt = bench.build(cfg, 1)
model.to_gpu(0)
opt.setup(model)
B = cfg['batch_per_gpu']
shards = [V.concat_examples(bench.synth_examples(B, cfg, seed=71 + s), device=0) for s in range(2)]
upd = V.VQVAE_ParallelUpdater(bench.ResidentIterator(shards), opt, comm=SingleCommunicator(),
                              converter=bench.resident_converter, device=0)
for _ in range(3):
    upd.update()
backend.synchronize()
names = ['', 'RESBLOCK_GATE', 'RESBLOCK_OUT', 'RESBLOCK_BWD_GZ', 'RESBLOCK_BWD_GX', 'RESBLOCK_BWD_GC', 'RESBLOCK_WGRAD',
         'CONV_FWD', 'CONV_BWD_DATA', 'CONV_WGRAD', 'VQ_NEAREST']
lib = _lib.load()
steps = 5
tot_all = 0.0
for tag in range(1, 11):
    lib.vqvae_prof_reset()
    lib.vqvae_prof_enable(1 << tag)
    for _ in range(steps):
        upd.update()
    backend.synchronize()
    lib.vqvae_prof_enable(0)
    tot, cnt = C.c_double(0), C.c_int(0)
    _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
    if cnt.value:
        print('%-18s %5.1f launches/step  avg %8.1f us  %6.2f ms/step' % (names[tag], cnt.value / steps, 1e3 * tot.value / cnt.value, tot.value / steps))
        tot_all += tot.value / steps
print('tagged total %.2f ms/step' % tot_all)
