#!/usr/bin/env python
"""BASELINE configs[3]: large-codebook VQ stress (k=8192, d=128).  Reports time, the
expansion-form FLOP rate (2*N*k*d) and the algorithmic HBM rate (4*(N*d + k*d + N + N*d))
of vqvae_vq_nearest_fwd at the training shape (N=1920) and a bandwidth-saturating N.
Dev tool: python tools/vq_stress.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))
from vqvae_amd import _lib, backend  # noqa: E402
from vqvae_amd.backend import DeviceArray  # noqa: E402


def inputs(B, d, T, k):
    rw = np.random.RandomState(2)
    W = (rw.standard_normal((k, d)) / np.sqrt(d)).astype(np.float32)
    rz = np.random.RandomState(1)
    N = B * T
    rows = rz.standard_normal((N, d)).astype(np.float32)
    j = rz.randint(0, k, size=N // 2)
    rows[N // 2:] = W[j] + np.float32(0.5) * rz.standard_normal((N - N // 2, d)).astype(np.float32)
    z = np.ascontiguousarray(rows.reshape(B, T, d).transpose(0, 2, 1))
    return z, W


def main():
    backend.init(0)
    d, T, k = 128, 120, 8192
    for B in (16, 1024, 8192):
        z, W = inputs(B, d, T, k)
        dz, dW = backend.to_device(z), backend.to_device(W)
        idx = DeviceArray((B, T), np.int32)
        e = DeviceArray((B, d, T), np.float32)
        nre = DeviceArray((1,), np.int32)
        ws = backend.workspace(_lib.load().vqvae_vq_workspace_bytes(B, d, T, k))
        def run():
            _lib.call('vqvae_vq_nearest_fwd', dz.ptr, dW.ptr, B, d, T, k, 0, idx.ptr, e.ptr, nre.ptr,
                      ws.ptr, ws.nbytes, backend.stream())
        run(); backend.synchronize()
        reps = 5 if B <= 1024 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        backend.synchronize()
        dt = (time.perf_counter() - t0) / reps
        N = B * T
        flop = 2.0 * N * k * d
        byts = 4.0 * (N * d + k * d + N + N * d)
        print('N=%8d  %9.3f ms  %7.1f TFLOP/s (expansion form)  %7.1f GB/s algorithmic  re-checked rows %d (%.2f%%)'
              % (N, dt * 1e3, flop / dt / 1e12, byts / dt / 1e9, int(nre.get()[0]), 100.0 * int(nre.get()[0]) / N))


if __name__ == '__main__':
    main()
