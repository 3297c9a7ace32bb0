"""Stage timing of the persistent generation kernel (library built with `make EXTRA=-DMEGA_PROF`)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))
from vqvae_amd import backend
from vqvae_amd.wavenet import WaveNet
from vqvae_amd.core import Variable
backend.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dec = WaveNet(2, 10, 2, 256, 256, 256, 256, 256, False, 30, -40, 192, 0)
dec.to_gpu()
dec(Variable(backend.zeros((1, 256, 2048, 1))), Variable(backend.zeros((1, 192, 2048, 1))))
steps = 4000
rs = np.random.RandomState(0)
cond = backend.to_device(rs.standard_normal((n, 192, steps + 1)).astype(np.float32))
u = rs.uniform(0.01, 0.99, (steps + 1, n))
dec.generate_sequence(cond, u, persistent=True)
ws = dec._gen._last_ws
prof = ws.flat_view(16, 32).get().view(np.int64)
names = ['embed + z_0 publish', 'shadow: gather x_l', 'shadow: u = Wc1 x_l', 'critical: gather z_l', 'x rows + z_{l+1} publish',
         'skip rows + weight hand-over', 'head: publish s + gather', 'proj1 + gather s1', 'proj2 + publish', 'sampler (wg 0)', 'feedback gather']
tot = prof[:11].sum()
for i, nm in enumerate(names):
    print('%-52s %8.2f us/step' % (nm, prof[i] * 0.01 / steps))
print('total %.1f us/step' % (tot * 0.01 / steps))
