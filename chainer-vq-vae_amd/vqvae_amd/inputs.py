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


class _Pinned(object):
    """A page-locked host buffer viewed as a NumPy array (source of asynchronous host -> device copies)."""

    def __init__(self, shape, dtype):
        import ctypes as C
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.call('vqvae_host_alloc', C.byref(p), self.nbytes)
        self.ptr = p.value
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=self.dtype).reshape(self.shape)

    def __del__(self):
        try:
            _lib.load().vqvae_host_free(self.ptr)
        except Exception:
            pass


class StreamingInputIterator(object):
    """The input leg of a real training step (updaters.py:8, 37-38; utils.py:85-110): every ``next()`` hands the
    updater a minibatch that was in HOST memory when the previous step started.

    ``source()`` returns ``(raw, speaker)`` -- float32 (B, L+1) peak-normalised crops and int (B,) speaker ids, i.e.
    Preprocess's output before the mu-law transform.  The crops (0.5 MB at B = 16 instead of the 125.8 MB one-hot)
    are copied into one of two page-locked buffers and sent to the device on a copy stream while the previous step's
    kernels run; the main stream waits for the copy's event, bins the waveform on the device (bit-exact with
    utils.py:18-23) and the step consumes x_enc / bin indices / targets exactly as from DeviceInputPipeline.
    Double buffering: set k's host and device buffers are reused two calls later -- the main stream's work on them
    (binning, encoder, loss) has been enqueued by then, and the copy that overwrites them is ordered behind it by an
    event recorded at hand-over."""

    yields_rank_shard = True

    class _Set(object):
        pass

    def __init__(self, source, batch, length, quantize=256):
        self.source, self.B, self.L1 = source, int(batch), int(length) + 1
        self.pipe = DeviceInputPipeline(quantize)
        self.copy_stream = backend.pool_stream(7)
        self.sets = []
        for _ in range(2):
            s = self._Set()
            s.h_raw = _Pinned((self.B, self.L1), np.float32)
            s.h_spk = _Pinned((self.B,), np.int32)
            s.d_raw = DeviceArray((self.B, 1, self.L1, 1), np.float32)
            s.d_spk = DeviceArray((self.B,), np.int32)
            s.ready = backend.Event()         # the copy has landed
            s.free = None                     # the main stream's readers of d_raw / d_spk are all enqueued (and this marks their end)
            s.submitted = False
            self.sets.append(s)
        self.i = 0
        self._submit(self.sets[0])

    def _submit(self, s):
        """Host side of one hand-over: fill set ``s``'s page-locked buffers, enqueue its copies on the copy stream."""
        if s.submitted:
            _lib.call('vqvae_event_synchronize', s.ready.h)      # the last copy FROM these host buffers has completed
        raw, spk = self.source()
        np.copyto(s.h_raw.array, np.asarray(raw, np.float32).reshape(self.B, self.L1))
        np.copyto(s.h_spk.array, np.asarray(spk, np.int32).reshape(self.B))
        if s.free is not None:
            backend.wait_event(self.copy_stream, s.free)         # ... and everything that read the device buffers
        _lib.call('vqvae_memcpy_h2d_async', s.d_raw.ptr, s.h_raw.ptr, s.h_raw.nbytes, self.copy_stream)
        _lib.call('vqvae_memcpy_h2d_async', s.d_spk.ptr, s.h_spk.ptr, s.h_spk.nbytes, self.copy_stream)
        s.ready.record(self.copy_stream)
        s.submitted = True

    def next(self):
        cur = self.sets[self.i & 1]
        self.i += 1
        main = backend.stream()
        backend.wait_event(main, cur.ready)
        B, L1 = self.B, self.L1
        # bins of the whole crop in one launch; quantized[:-1] -> decoder input (indices), quantized[1:] -> targets
        q_all = self.pipe.bins_device(cur.d_raw.reshape(B, L1))
        x_dec = IndexInput((B, L1 - 1), self.pipe.quantize)
        t = DeviceArray((B, L1 - 1, 1), np.int32)
        _shift_rows(q_all, x_dec, 0)
        _shift_rows(q_all, t, 1)
        x_enc, spk = cur.d_raw.copy(), cur.d_spk.copy()          # (the step's Variables may outlive the set's turn)
        cur.free = backend.Event().record(main)                  # every reader of cur's device buffers is enqueued
        self._submit(self.sets[self.i & 1])                      # the NEXT batch travels under this step's kernels
        return self.Batch((x_enc, x_dec, spk, t))

    class Batch(object):
        def __init__(self, arrays):
            self.arrays = arrays


def _shift_rows(q_all, out, shift):
    """out[b, :] = q_all[b, shift : shift + L] (int32 rows of L+1 -> rows of L): one device-to-device 2-D copy."""
    B, L1 = q_all.shape
    L = L1 - 1
    _lib.call('vqvae_memcpy2d_d2d', out.ptr, 4 * L, q_all.ptr + 4 * shift, 4 * L1, 4 * L, B, backend.stream())
