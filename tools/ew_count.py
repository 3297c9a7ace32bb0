import os, sys, collections
ROOT = '/root/repo'
sys.path[:0] = [os.path.join(ROOT, 'chainer-vq-vae_amd'), ROOT]
import numpy as np
import bench
import vqvae_amd as V
from vqvae_amd import _lib, backend
backend.init(0)
cfg = dict(bench.CFG)
model, opt = bench.build(cfg, 1)
model.to_gpu(0); opt.setup(model)
ex = bench.synth_examples(16, cfg, seed=71)
shard = V.concat_examples(ex, device=0)
it = bench.ResidentIterator([shard])
upd = V.VQVAE_ParallelUpdater(it, opt, converter=bench.resident_converter, device=0)
for _ in range(2): upd.update()
backend.synchronize()
orig = _lib.call
cnt = collections.Counter(); byt = collections.Counter()
names = ['ADD','SUB','MUL','AXPBY','SCALE','SQUARE','RELU','RELU_BWD','FILL','MUL_SCALAR_DEV']
import traceback
where = collections.Counter()
def call(name, *a):
    if name == 'vqvae_elementwise':
        k = (names[a[0]], int(a[1]))
        cnt[k] += 1
        st = traceback.extract_stack(limit=6)
        where[(names[a[0]], int(a[1]), st[-3].name + '<' + st[-4].name)] += 1
    return orig(name, *a)
_lib.call = call
import vqvae_amd.functions as F, vqvae_amd.core as core
for m in (F, core):
    if hasattr(m, '_lib'): pass
upd.update()
backend.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][1] * kv[1]):
    print(k, v)
print(sum(cnt.values()))
for k, v in sorted(where.items(), key=lambda kv: -kv[0][1] * kv[1])[:14]:
    print(k, v)
