"""Weight slabs of the generic convolutions, packed AHEAD of the launches that read them.

``vqvae_conv1d_fwd`` / ``_bwd_data`` re-lay W into their workspace in front of every GEMM (pack_kernel, for a large
'float32x2' launch also wamax_kernel).  For the latency-bound chains of small convs -- the encoder, the condition embed and
their backward, proj1 / proj2 -- those ~35 little launches sit on the step's critical path, a pack in front of every conv.
The weights only change in the optimizer, so a step can pack ALL its slabs at once when it starts: on the side stream,
a few batched launches (vqvae_conv1d_pack takes 24 jobs per launch) beside the first kernels of the forward, and every conv
of the step then finds its slab ready (vqvae_conv1d_amax::packed).

There is no model-specific list: a step records which (parameter, conv geometry, direction) triples it packed, and the
NEXT step's ``prefetch()`` (called by ``VAE.__call__`` before anything else) packs exactly those, in their order of use.  A
slab is handed out only while it provably matches what the launch would pack itself: same parameter buffer, same parameter
version (optimizer step, load_npz), same matmul mode and float32x2 threshold -- anything else packs in line as before.
Only ``Parameter`` weights are prefetched (a weight computed inside the step does not exist yet when the step starts).
``prepack.ENABLED = False`` turns it off (bitwise the same step: the same pack kernels write the same slabs)."""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib, backend, core
from .backend import DeviceArray

ENABLED = True
_MAX_JOBS = 24            # MAXSEG of csrc/gemm_common.h: jobs of one pack launch
_FIRST_JOBS = 6            # jobs of a prefetch's first launch

_used = {}                # key -> (weight Variable, desc, backward): this step's convs in order of first use (the next plan)
_ready = {}               # key -> [packed DeviceArray, launch = [Event, main stream has waited], version]: one event per batched launch
_last = None              # the last launch of the last prefetch (every earlier one is in front of it on the side stream)
_joined = True
stats = {'hits': 0, 'misses': 0}      # lookups served from a prefetched slab / packed in line (tests)


def _desc_key(d):
    return (d.B, d.Cin, d.Tin, d.Cout, d.Tout, d.K, d.stride, d.pad, d.dil)      # (relu does not touch the weights)


def _version(wv, desc):
    """Everything that can change the VALUES behind a parameter's pointer or the slab a launch would pack from them: the
    optimizer's step count (Adam / EMA kernels), the load / layout / init epochs (serializers, arena adoption, lazily shaped
    parameters), the memory's value version (in-place writes through DeviceArray methods -- Link.copyparams, p.data.set(...) --
    and a recycled address: backend._Block.wver), the matmul mode and the float32x2 threshold."""
    step = getattr(wv, '_owner_step', None)
    lib = _lib.load()
    W = wv.data
    return (step() if step is not None else 0, core.param_epoch('load'), core.param_epoch('layout'), core.param_epoch('init'),
            W.wver if isinstance(W, DeviceArray) else -1,
            lib.vqvae_get_matmul_dtype(), lib.vqvae_conv1d_uses_f32x2(C.byref(desc)))


def lookup(wv, W, desc, backward):
    """The slab vqvae_conv1d_fwd* (backward = 0) / _bwd_data* (1) would pack from ``W`` for ``desc``, if this step's
    prefetch packed it: its device pointer (the main stream is behind the pack from here on), else None.  Either way
    the use is recorded for the next step's prefetch."""
    if not ENABLED or not isinstance(wv, core.Parameter) or not isinstance(W, DeviceArray):
        return None
    key = (W.ptr, int(backward), _desc_key(desc))
    if key not in _used:
        d = _lib.Conv1dDesc()
        C.pointer(d)[0] = desc
        _used[key] = (weakref.ref(wv), d, int(backward))      # (weak: a plan must not keep a dead model's parameters -- and their device buffers -- alive)
    e = _ready.get(key)
    if e is None or e[2] != _version(wv, desc):
        stats['misses'] += 1
        return None
    stats['hits'] += 1
    launch = e[1]
    if not launch[1]:                        # the first reader of a launch's slabs puts the main stream behind the launch
        backend.wait_event(backend.stream(), launch[0])
        launch[1] = True
    return e[0].ptr


def prefetch():
    """Packs, on the side stream, every slab the previous step looked up (VAE.__call__ calls this first)."""
    global _used, _ready, _last, _joined
    plan, _used = list(_used.values()), {}
    join()                                   # (a slab nobody waited for: the main stream gets behind it before it is freed)
    _ready = {}
    if not ENABLED or not plan:
        return
    lib = _lib.load()
    jobs = []
    for wr, desc, backward in plan:
        wv = wr()
        if wv is None:                       # the model is gone
            continue
        W = wv.data
        if not isinstance(W, DeviceArray) or W.ndim < 3 or (W.shape[0], W.shape[1], W.shape[2]) != (desc.Cout, desc.Cin, desc.K):
            continue
        nbytes = lib.vqvae_conv1d_packed_bytes(C.byref(desc), backward)
        if nbytes == 0:
            continue
        jobs.append((wv, W, desc, backward, DeviceArray((int(nbytes) // 4,), np.float32)))
    if not jobs:
        return
    main, side = backend.stream(), backend.side_stream()
    backend.wait_event(side, backend.Event().record(main))     # the optimizer's last write of the parameters; the buffers' last readers
    # (the step's first convs read the first launch's slabs: a short one, so that they do not wait for everything)
    cuts = [0, min(_FIRST_JOBS, len(jobs))]
    while cuts[-1] < len(jobs):
        cuts.append(min(cuts[-1] + _MAX_JOBS, len(jobs)))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        chunk = jobs[lo:hi]
        n = len(chunk)
        descs = (_lib.Conv1dDesc * n)()
        for i, j in enumerate(chunk):
            C.pointer(descs[i])[0] = j[2]
        Ws = (C.c_void_p * n)(*[j[1].ptr for j in chunk])
        bw = (C.c_int * n)(*[j[3] for j in chunk])
        pk = (C.c_void_p * n)(*[j[4].ptr for j in chunk])
        _lib.call('vqvae_conv1d_pack', n, descs, Ws, bw, pk, side)
        launch = [backend.Event().record(side), False]
        for wv, W, desc, backward, buf in chunk:
            _ready[(W.ptr, backward, _desc_key(desc))] = [buf, launch, _version(wv, desc)]
        _last = launch
    _joined = False


def join():
    """The main stream waits for the last prefetch as a whole (no-op once done): VAE.__call__ ends with it, so that the
    side stream's packs are joined inside the step whether or not every slab found its reader (a recorded step must not
    end with unjoined work; a slab must not be freed with its pack formally outstanding)."""
    global _joined
    if not _joined and _last is not None:
        if not _last[1]:
            backend.wait_event(backend.stream(), _last[0])
        for e in _ready.values():
            e[1][1] = True
    _joined = True


def reset():
    """Forgets the plan and the slabs (tests)."""
    global _used, _ready
    join()
    _used, _ready = {}, {}
    stats['hits'] = stats['misses'] = 0
