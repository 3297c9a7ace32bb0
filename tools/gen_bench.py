"""Throughput of the device-resident generation loop (generate.py:105-145) at BASELINE decoder sizes.
usage: python tools/gen_bench.py [--workload c2|c5] [--n 1] [--steps 16000] [--graph-steps 8]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2')
    ap.add_argument('--n', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16000)
    ap.add_argument('--graph-steps', type=int, default=8)
    ap.add_argument('--group', type=int, default=4)
    ap.add_argument('--streams', type=int, default=8)
    ap.add_argument('--batch', type=int, default=0, help='N sequences through generate_batch (concurrent groups of 4)')
    ap.add_argument('--per-step-kernels', action='store_true', help='hipGraph of per-step kernels instead of the persistent kernel')
    a = ap.parse_args()
    from vqvae_amd import backend
    from vqvae_amd.wavenet import WaveNet
    backend.init(0)
    mol = a.workload == 'c5'
    n_loop = 4 if mol else 2
    dec = WaveNet(n_loop, 10, 2, 1 if mol else 256, 256, 256, 256, 256, mol, 30, -40, 192, 0)
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    # materialise the lazily initialised parameters with one tiny forward
    x = backend.zeros((1, 1 if mol else 256, 2048, 1))
    c = backend.zeros((1, 192, 2048, 1))
    dec.to_gpu()
    dec(Variable(x), Variable(c))
    T = a.steps + 1
    if a.batch:
        rs = np.random.RandomState(0)
        cond = backend.to_device(rs.standard_normal((a.batch, 192, T)).astype(np.float32))
        u = rs.uniform(0.01, 0.99, (T, a.batch, 10 if mol else 1))
        dec.generate_batch(cond, u, n_steps=64, group=a.group, max_streams=a.streams)
        t0 = time.time()
        o = dec.generate_batch(cond, u, group=a.group, max_streams=a.streams)
        dt = time.time() - t0
        print('group %d streams %d: ' % (a.group, a.streams) + 'batch of %d sequences x %d steps in %.3f s = %.0f samples/s aggregate (%.1f sequences in real time at 16 kHz)'
              % (a.batch, a.steps, dt, a.batch * a.steps / dt, a.batch * a.steps / dt / 16000))
        return
    rs = np.random.RandomState(0)
    cond = backend.to_device(rs.standard_normal((a.n, 192, T)).astype(np.float32))
    u = rs.uniform(0.01, 0.99, (T, a.n, 10 if mol else 1))
    pers = not a.per_step_kernels
    dec.generate_sequence(cond, u, n_steps=64, graph_steps=a.graph_steps, persistent=pers)      # warm-up
    t0 = time.time()
    out = dec.generate_sequence(cond, u, graph_steps=a.graph_steps, persistent=pers)
    dt = time.time() - t0
    o = out.get()
    print('%s workload %s n=%d: %d steps in %.3f s = %.1f us/step, %.0f samples/s per sequence (%.2fx real time at 16 kHz), '
          'distinct outputs %d' % ('persistent' if pers else 'per-step kernels', a.workload, a.n, a.steps, dt, 1e6 * dt / a.steps, a.steps / dt,
                                   a.steps / dt / 16000, len(np.unique(o))))


if __name__ == '__main__':
    main()
