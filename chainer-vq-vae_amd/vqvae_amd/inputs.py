"""Device-side input pipeline: the step right before the hot path.

The reference's ``Preprocess`` (utils.py:54-110) builds, per example, a float one-hot of the
mu-law bins -- 125.8 MB per 16-example minibatch that ``converter`` then copies to the GPU every
step (updaters.py:8).  Here the host ships only the normalised waveform crops (0.5 MB); the
GPU bins them (bit-exact with utils.py:18-23) and the decoder's embed conv consumes the bin
INDICES directly (a 2-column gather, bit-identical to the dense conv on the one-hot tensor).
"""
import random

import numpy as np

from . import _lib, backend
from .backend import DeviceArray
from .utils import MuLaw


def _f32_key(x):
    """Order-preserving int64 key of float32 values."""
    b = np.asarray(x, np.float32).view(np.int32).astype(np.int64)
    return np.where(b < 0, -(b & 0x7fffffff), b)


def _key_f32(k):
    k = np.asarray(k, np.int64)
    b = np.where(k < 0, (-k) | 0x80000000, k).astype(np.uint32)
    return b.view(np.float32)


def mulaw_thresholds(mu=256):
    """thr[j-1], j = 1..mu-1: the smallest float32 x in [-1, 1] with MuLaw(mu).transform(x) >= j,
    by bisection over the ordered float32 values against the NumPy transform itself.  The
    transform is monotone, hence bin(x) = #{j : x >= thr[j-1]} for every x in [-1, 1]."""
    f = MuLaw(mu).transform
    j = np.arange(1, mu)
    lo = np.full(j.shape, _f32_key(np.float32(-1.0)), np.int64)     # f(lo) < j
    hi = np.full(j.shape, _f32_key(np.float32(1.0)), np.int64)      # f(hi) >= j
    while np.any(hi - lo > 1):
        mid = (lo + hi) // 2
        ge = f(_key_f32(mid)) >= j
        hi = np.where(ge, mid, hi)
        lo = np.where(ge, lo, mid)
    return _key_f32(hi).astype(np.float32)


def crop_or_pad(wave, length, start=None, rng=random):
    """The padding / trimming branch of Preprocess.__call__ (utils.py:57-58, 65-81) for one loaded
    and silence-trimmed waveform: peak-normalise, then zero-pad to ``length + 1`` samples or take
    ``length + 1`` samples from ``start`` (default: ``rng.randint(0, len - (length+1) - 1)``, the
    reference's unseeded ``random``).  Zero padding of the waveform IS the reference's padding of
    the bins with quantize // 2, because MuLaw.transform(0.0) == quantize // 2 (utils.py:74)."""
    L = length + 1                                                 # utils.py:47
    raw = np.asarray(wave, np.float32)
    raw = (raw / np.abs(raw).max()).astype(np.float32)             # utils.py:58-59
    if len(raw) <= L:
        return np.concatenate((raw, np.zeros(L - len(raw), np.float32)))
    if start is None:
        start = rng.randint(0, len(raw) - L - 1)                   # utils.py:78
    return raw[start:start + L]


class IndexInput(DeviceArray):
    """(B, T) int32 bin indices standing in for the (B, q, T, 1) one-hot decoder input."""
    __slots__ = ('quantize',)

    def __init__(self, shape, quantize):
        DeviceArray.__init__(self, shape, np.int32)
        self.quantize = quantize


class DeviceInputPipeline(object):
    """raw crops -> (x_enc, x_dec, speaker, t) on the device, with x_dec as bin indices.

    ``raw``: float32 (B, L+1) peak-normalised crops in [-1, 1] (what utils.py:58-81 produces);
    returns the same 4-tuple contract as Preprocess + converter, except that x_dec is an
    ``IndexInput`` (B, L) instead of a one-hot (B, q, L, 1) tensor."""

    def __init__(self, quantize=256):
        self.quantize = quantize
        self._thr_host = mulaw_thresholds(quantize)
        self._thr = None

    def bins(self, host_f32):
        if self._thr is None:
            self._thr = backend.to_device(self._thr_host)
        x = backend.to_device(np.ascontiguousarray(host_f32, np.float32))
        return self.bins_device(x)

    def bins_device(self, x, out=None):
        if self._thr is None:
            self._thr = backend.to_device(self._thr_host)
        q = out if out is not None else DeviceArray(x.shape, np.int32)
        _lib.call('vqvae_mulaw_bins', x.ptr, x.size, self._thr.ptr, self._thr.size, q.ptr,
                  backend.stream())
        return q

    def from_waveforms(self, waves, speaker, length, starts=None, rng=random):
        """Variable-length waveforms -> minibatch: crop_or_pad each (host, a few KB), then bin on
        the device.  ``starts``: optional explicit crop offsets (None entries draw from ``rng``)."""
        starts = [None] * len(waves) if starts is None else starts
        raw = np.stack([crop_or_pad(w, length, s, rng) for w, s in zip(waves, starts)])
        return self(raw, speaker)

    def __call__(self, raw, speaker):
        raw = np.ascontiguousarray(raw, np.float32)
        B, L1 = raw.shape
        x_enc = backend.to_device(raw.reshape(B, 1, L1, 1))
        x_dec = IndexInput((B, L1 - 1), self.quantize)
        self.bins_device(backend.to_device(raw[:, :-1]), out=x_dec)        # quantized[:-1]
        t = self.bins_device(backend.to_device(raw[:, 1:])).reshape(B, L1 - 1, 1)   # quantized[1:]
        spk = backend.to_device(np.asarray(speaker, np.int32))
        return x_enc, x_dec, spk, t
