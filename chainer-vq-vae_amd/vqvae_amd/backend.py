"""Device arrays, the caching allocator and the per-process stream.

This is the plumbing Chainer gets from CuPy (ndarray + memory pool + streams);
here it sits directly on the C ABI (vqvae_malloc / vqvae_memcpy_* /
vqvae_stream_*).  One process drives one GPU on one stream; every kernel of the
library is enqueued on ``stream()``.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib

_state = {'stream': None, 'side': None, 'device': None, 'pool': {}, 'live_bytes': 0,
          'pool_bytes': 0, 'ws': {}, 'events': [], 'overlap': False}


def init(device=0):
    """Selects the GPU and creates the process stream (idempotent)."""
    if _state['stream'] is not None:
        if device != _state['device']:
            raise RuntimeError('backend already initialised on device %d' % _state['device'])
        return
    lib = _lib.load()
    n = C.c_int(0)
    rc = lib.vqvae_device_count(C.byref(n))
    if rc != 0 or n.value < 1:
        raise RuntimeError(
            'no HIP device visible (%s): this framework has no CPU compute path'
            % lib.vqvae_last_error_string().decode())
    _lib.call('vqvae_set_device', device)
    s = C.c_void_p()
    _lib.call('vqvae_stream_create', C.byref(s))
    _state['stream'] = s
    _state['device'] = device
    if not _state.get('matmul_explicit'):          # a mode chosen before init() stays
        _set_matmul_code(_MATMUL_CODES[default_matmul_dtype()])


def available():
    """True when the library loads and at least one GPU is visible."""
    try:
        lib = _lib.load()
    except (ImportError, OSError):
        return False
    n = C.c_int(0)
    return lib.vqvae_device_count(C.byref(n)) == 0 and n.value > 0


def stream():
    if _state['stream'] is None:
        init(0)
    return _state['stream']


def side_stream():
    """Second HIP stream for work that is independent of the serial backward/forward chain
    (weight gradients, the skip GEMM): its workgroups fill the residency tail of the chain's
    kernels.  Ordering is by events only (record_event / wait_event)."""
    if _state['side'] is None:
        stream()
        s = C.c_void_p()
        _lib.call('vqvae_stream_create', C.byref(s))
        _state['side'] = s
    return _state['side']


def pool_stream(i):
    """i-th stream of a small pool for independent concurrent launches (batched generation)."""
    pool = _state.setdefault('pool_streams', [])
    while len(pool) <= i:
        stream()
        s = C.c_void_p()
        _lib.call('vqvae_stream_create', C.byref(s))
        pool.append(s)
    return pool[i]


def overlap_enabled():
    """Whether backward runs weight gradients on the side stream.  Off by default: with the
    float32x3 kernels (one 8-wave workgroup per CU, the chip at its power limit) a second stream
    only interleaves work that cannot co-reside -- measured +0.58 ms per step at configs[1]; it
    was worth -1.6 ms with round 1's fp32 MFMA kernels and is neutral in that mode now.
    set_overlap(True) turns it on."""
    return _state['overlap']


def set_overlap(on):
    _state['overlap'] = bool(on)


class Event(object):
    """A pooled hipEvent."""

    def __init__(self):
        if _state['events']:
            self.h = _state['events'].pop()
        else:
            e = C.c_void_p()
            _lib.call('vqvae_event_create', C.byref(e))
            self.h = e

    def record(self, s):
        _lib.call('vqvae_event_record', self.h, s)
        return self

    def __del__(self):
        try:
            _state['events'].append(self.h)
        except Exception:
            pass


def wait_event(s, event):
    """Stream ``s`` waits on the GPU until ``event`` has completed."""
    _lib.call('vqvae_stream_wait_event', s, event.h)


_MATMUL_CODES = {'float32': 0, 'fp32': 0, 'bfloat16': 1, 'bf16': 1, 'float32x3': 2, 'fp32x3': 2,
                 'float32x2': 3, 'fp32x2': 3}


def default_matmul_dtype():
    """The matmul mode a process starts in: $VQVAE_MATMUL if set, else 'float32x2' (fp32 products as three
    fp16 MFMA products; as accurate as 'float32', the fp32 MFMA path, and 1.85x faster; 'float32x3', six bf16
    products of an exact split, was the default until round 4)."""
    name = os.environ.get('VQVAE_MATMUL', 'float32x2')
    if name not in _MATMUL_CODES:
        raise ValueError('VQVAE_MATMUL=%r: expected one of %s' % (name, sorted(_MATMUL_CODES)))
    return name


def _set_matmul_code(code):
    _lib.call('vqvae_set_matmul_dtype', code)


def set_matmul_dtype(name):
    """'float32' (fp32 MFMA), 'bfloat16' (operands rounded to bf16, fp32 accumulate),
    'float32x3' (fp32 products as six bf16 MFMA products of an exact three-way operand split:
    fp32 accuracy at 0.375 of the fp32 MFMA time -- csrc/gemm_common.h, "matmul mode 2") or
    'float32x2' (the default: three fp16 MFMA products of a two-piece split of operands scaled by a power of
    two per tensor -- "matmul mode 3": fp32 accuracy at 0.19 of the fp32 MFMA time; the tensors' absolute maxima
    travel with them through ResidualNet's chain, other large convs scan their operand once).
    An explicit choice survives a later backend.init(); workspace sizes depend on the mode, so
    choose it before building workspaces.

    Non-finite operands.  'float32x3' splits x into bf16(x) + bf16(x - bf16(x)) + ...: an operand
    that is +-Inf gives Inf - Inf = NaN in the remainder, and a finite |x| above ~3.39e38 rounds its
    high piece to Inf, so both produce NaN where the fp32 MFMA path gives +-Inf or a finite value.
    The path's tensors (audio in [-1, 1], LeCun-scaled weights, their gradients) stay far from that
    range; a caller that needs IEEE Inf propagation through a conv should use 'float32'
    (tests/test_gpu_kernels.py::test_float32x3_nonfinite_operands pins the behaviour)."""
    if name not in _MATMUL_CODES:
        raise ValueError('set_matmul_dtype(%r): expected one of %s' % (name, sorted(_MATMUL_CODES)))
    _state['matmul_explicit'] = True
    _set_matmul_code(_MATMUL_CODES[name])


def set_f32x2_min_gflop(gflop):
    """'float32x2' only: the smallest generic conv launch (in GFLOP) that runs the three-product kernels -- they need
    one extra pass over the operand for its absolute maximum; default 8 (proj1 / proj2 at the configs), 0 = every
    launch (how the tests reach those kernels at their small shapes).  ResidualNet's chain is not affected: its
    maxima travel with the tensors."""
    _lib.call('vqvae_set_f32x2_min_gflop', float(gflop))


def defer_to_side(keep, writes=None):
    """Work has been enqueued on the side stream that the main stream does not wait for yet; ``keep``: everything that work
    reads or writes through temporaries (the allocator invariant below: a block must not return to the pool before a join
    has been enqueued behind its last side-stream user).  ``writes``: the arrays that work WRITES (gradient slots): only an
    accumulation into one of those has to join early (side_writes_pending); None = unknown, every accumulation joins.
    join_side() closes the window."""
    _state.setdefault('side_keep', []).append(keep)
    _state['side_pending'] = True
    if writes is None:
        _state['side_writes'] = None
    elif _state.get('side_writes', set()) is not None:
        _state.setdefault('side_writes', set()).update(int(w.ptr) for w in writes if w is not None)


def side_writes_pending(*arrays):
    """True when work deferred to the side stream may still be writing one of ``arrays`` (by device pointer): the main stream
    must join before it reads or accumulates into it.  A gradient that no deferred launch touches -- every parameter of the
    encoder / condition embed / quantiser while the decoder's last weight gradients run beside their backward -- does not."""
    if not _state.get('side_pending'):
        return False
    w = _state.get('side_writes', set())
    if w is None:
        return True
    return any(a is not None and int(a.ptr) in w for a in arrays)


class deferred_join(object):
    """While open, backward sweeps do not join the side stream when they end (core.Variable.backward): the updaters run
    their two or three sweeps inside one (updaters.three_loss_backward), so that the codebook loss's small sweep also
    runs beside the deferred weight gradients; closing it joins.  Host reads (DeviceArray.get) join on their own."""

    def __enter__(self):
        _state['lazy_join'] = _state.get('lazy_join', 0) + 1

    def __exit__(self, *exc):
        _state['lazy_join'] -= 1
        if _state['lazy_join'] == 0:
            join_side()
        return False


def join_side(force=True):
    """The main stream waits for everything deferred to the side stream (no-op when nothing is pending).  Called when a
    backward sweep ends (core.Variable.backward; force=False: not inside a deferred_join), before anything reads the
    gradients."""
    if not force and _state.get('lazy_join', 0) > 0:
        return
    if _state.get('side_pending'):
        wait_event(stream(), Event().record(side_stream()))
        _state['side_pending'] = False
        _state['side_keep'] = []
        _state['side_writes'] = set()


def set_presplit(mask):
    """'float32x2' only: which tensors of ResidualNet's chain are kept PRE-SPLIT in HBM (fp16 hi | lo dwords written once
    by their producer, csrc/gemm_common.h "PRE-SPLIT storage"): bit 0 = gh_l, bit 1 = the residual stream x_l; bit 2: the
    gate kernel saves sigmoid and z only (the backward takes tanh = z / sigmoid: VQVAE_STORE_GATES_SIG); default 7
    0 = every tensor fp32 and every reader splits for itself (round 4's form, the A/B alternate)."""
    _lib.call('vqvae_set_presplit', int(mask))


# --------------------------------------------------------------------------- #
# The run-time guard of 'float32x2' ("the dynamic-range contract").  ResidualNet's chain keeps x_l and gh_l
# PRE-SPLIT under a-priori bounds; a bound 2^m above the tensor's actual maximum costs m bits of the mode's 2^-39 absolute
# floor.  Every backward sweep ends with one tiny launch (vqvae_f32x2_contract_check) that compares, on the device, each
# pre-split tensor's bound with the maximum its producer published and counts the tensors beyond 2^CONTRACT_LOG2_LIMIT.
# Nothing is read back unless somebody asks:
#   f32x2_contract_violations()         -> {'violations', 'checked', 'worst_log2'} since the last reset (synchronises)
#   set_contract_debug(True): the chain reads the report after every sweep and RAISES on a violation
# A violation means: switch to 'float32x3' (no scales) or withdraw the pre-split storage (set_presplit(4)).
# --------------------------------------------------------------------------- #
CONTRACT_LOG2_LIMIT = 8


def contract_report():
    """The three device words (violations, float bits of the worst bound / max, tensors checked); allocated OUTSIDE any
    recording's arena, once, so that eager and replayed steps count into the same words."""
    r = _state.get('contract')
    if r is None:
        arena, _state['arena'] = _state.get('arena'), None
        try:
            r = zeros((4,), np.uint32)
        finally:
            _state['arena'] = arena
        _state['contract'] = r
    return r


def contract_debug():
    return bool(_state.get('contract_debug', False))


def set_contract_debug(on):
    _state['contract_debug'] = bool(on)


def f32x2_contract_violations(reset=False):
    """How many pre-split tensors of the 'float32x2' chain were split under a bound more than 2^CONTRACT_LOG2_LIMIT above
    their actual maximum since the last reset (0 = the mode delivered what DESIGN.md 3a promises on every step so far)."""
    if _state.get('contract') is None:
        return {'violations': 0, 'checked': 0, 'worst_log2': None, 'log2_limit': CONTRACT_LOG2_LIMIT}
    synchronize()
    w = _state['contract'].get()
    worst = float(np.array([w[1]], np.uint32).view(np.float32)[0])
    out = {'violations': int(w[0]), 'checked': int(w[2]),
           'worst_log2': (float(np.log2(worst)) if worst > 0 else None), 'log2_limit': CONTRACT_LOG2_LIMIT}
    if reset:
        _lib.call('vqvae_memset', _state['contract'].ptr, 0, 16, stream())
        _state['contract_seen'] = 0
    return out


def synchronize():
    _lib.call('vqvae_stream_synchronize', stream())
    if _state['side'] is not None:
        _lib.call('vqvae_stream_synchronize', _state['side'])
    for s in _state.get('pool_streams', []):
        _lib.call('vqvae_stream_synchronize', s)


def device_info():
    name = C.create_string_buffer(256)
    ncu = C.c_int(0)
    mem = C.c_size_t(0)
    stream()
    _lib.call('vqvae_device_info', name, 256, C.byref(ncu), C.byref(mem))
    return {'name': name.value.decode(), 'n_cu': ncu.value, 'total_mem': mem.value}


# --------------------------------------------------------------------------- #
# caching allocator: exact-size free lists (a training step repeats its sizes)
#
# INVARIANT (stream safety).  A block returns to the pool when the last Python reference to it
# drops -- not when the GPU work that uses it has finished.  That is safe on ONE stream (a later
# user of the block is enqueued behind the earlier one).  Work on the side / pool streams must
# therefore keep every operand referenced until a join event on the main stream has been
# enqueued AFTER it: ResidualStackFunction.backward holds gh / saved activations / scratch in
# locals until its final wait_event; generation.run_many synchronises around its launches.  A
# new side-stream use that frees a temporary earlier would corrupt memory silently.
# --------------------------------------------------------------------------- #
def _round(nbytes):
    return max(256, (nbytes + 255) // 256 * 256)


class Arena(object):
    """The memory of ONE captured launch sequence (a training step recorded into a hipGraph, updaters.GraphedStep).
    While the arena is active every new block is served from the arena's own free lists (else from the pool) and
    belongs to the arena from then on: when its last reference drops it returns to the ARENA, never to the pool, so
    the addresses the recorded kernels were given can only be reused by the recording itself -- in the same order at
    every replay -- and by nothing else, for as long as the arena lives.  ``release()`` hands everything back."""

    def __init__(self):
        self.free = {}           # nbytes -> [ptr]: blocks of this arena nobody references right now
        self.alive = True

    def release(self):
        """Idle blocks go back to the pool now, blocks still referenced when their last reference drops."""
        if not self.alive:
            return
        self.alive = False
        for nbytes, ptrs in self.free.items():
            _state['pool'].setdefault(nbytes, []).extend(ptrs)
            _state['pool_bytes'] += nbytes * len(ptrs)
        self.free = {}


def arena_begin(arena):
    if _state.get('arena') is not None:
        raise RuntimeError('an allocation arena is already active')
    _state['arena'] = arena


def arena_end():
    _state['arena'] = None


_wtick = [0]


def _tick():
    """A process-unique, increasing stamp (see _Block.wver)."""
    _wtick[0] += 1
    return _wtick[0]


class _Block(object):
    # wver: the VALUE version of the memory -- a fresh stamp at allocation and at every in-place write that goes through a
    # DeviceArray method (set / copy_from / fill_zero: Link.copyparams, p.data.set(...), manual edits); views share it.
    # Kernels that write parameters (Adam, EMA, arena adoption, load_npz) are covered by the optimizer's step count and
    # core.param_epoch(...).  What caches of derived data (packed weight slabs: prepack.py, wavenet._pack_key) key on
    # besides the pointer, so that neither an edit nor a recycled address can serve slabs packed from other values.
    __slots__ = ('ptr', 'nbytes', 'arena', 'wver', '__weakref__')

    def __init__(self, nbytes):
        self.wver = _tick()
        nbytes = _round(nbytes)
        arena = _state.get('arena')
        self.arena = arena
        afree = arena.free.get(nbytes) if arena is not None else None
        if afree:
            self.ptr = afree.pop()
        else:
            free = _state['pool'].get(nbytes)
            if free:
                self.ptr = free.pop()
                _state['pool_bytes'] -= nbytes
            else:
                stream()
                p = C.c_void_p()
                _lib.call('vqvae_malloc', C.byref(p), nbytes)
                self.ptr = p.value
        self.nbytes = nbytes
        _state['live_bytes'] += nbytes

    def __del__(self):
        try:
            _state['live_bytes'] -= self.nbytes
            if self.arena is not None and self.arena.alive:
                self.arena.free.setdefault(self.nbytes, []).append(self.ptr)
                return
            _state['pool'].setdefault(self.nbytes, []).append(self.ptr)
            _state['pool_bytes'] += self.nbytes
        except Exception:      # interpreter shutdown
            pass


def free_all_blocks():
    """Returns every cached (unused) block to the driver."""
    for nbytes, ptrs in _state['pool'].items():
        for p in ptrs:
            _lib.call('vqvae_free', p)
    _state['pool'].clear()
    _state['pool_bytes'] = 0


def memory_stats():
    return {'live_bytes': _state['live_bytes'], 'cached_bytes': _state['pool_bytes']}


class DeviceArray(object):
    """A contiguous device buffer with NumPy-like metadata (the ``xp.ndarray``
    of this backend).  Views share the owning block."""
    __slots__ = ('ptr', 'shape', 'dtype', '_block', 'amax', 'relu_out', 'relu_masked', '__weakref__')

    def __init__(self, shape, dtype=np.float32, _block=None, _ptr=None):
        # matmul mode 'float32x2': the tensor's absolute maximum (an upper bound will do), when a producer published
        # it -- a DeviceArray of _lib.AMAX_SLOTS uint32 (see vqvae_absmax); travels with views, dropped by writes
        self.amax = None
        # relu_out: these values are the output of a ReLU (a conv that reads them may apply that ReLU's backward to the
        # gradient it produces: vqvae_conv1d_bwd_data_relu); relu_masked: this GRADIENT already carries that mask
        self.relu_out = False
        self.relu_masked = False
        if isinstance(shape, int):
            shape = (shape,)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        if _block is None:
            _block = _Block(self.nbytes)
            _ptr = _block.ptr
        self._block = _block
        self.ptr = _ptr

    # ---- metadata ----
    @property
    def wver(self):
        """Value version of the underlying memory (see _Block.wver)."""
        return self._block.wver

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return 'DeviceArray(shape=%s, dtype=%s, ptr=0x%x)' % (self.shape, self.dtype, self.ptr)

    # ---- views ----
    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            i = shape.index(-1)
            known = 1
            for j, s in enumerate(shape):
                if j != i:
                    known *= s
            shape[i] = self.size // known
        n = 1
        for s in shape:
            n *= s
        if n != self.size:
            raise ValueError('cannot reshape %s into %s' % (self.shape, tuple(shape)))
        out = DeviceArray(tuple(shape), self.dtype, _block=self._block, _ptr=self.ptr)
        out.amax = self.amax
        out.relu_out, out.relu_masked = self.relu_out, self.relu_masked
        return out

    def flat_view(self, offset, size, shape=None):
        """View of ``size`` elements starting ``offset`` elements in."""
        if offset < 0 or offset + size > self.size:
            raise ValueError('flat_view out of range')
        return DeviceArray(shape if shape is not None else (size,), self.dtype,
                           _block=self._block, _ptr=self.ptr + offset * self.dtype.itemsize)

    # ---- transfers ----
    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError('size mismatch in DeviceArray.set: %s vs %s' % (host.shape, self.shape))
        _lib.call('vqvae_memcpy_h2d', self.ptr, host.ctypes.data, self.nbytes, stream())
        self.amax = None
        self._block.wver = _tick()
        return self

    def get(self):
        join_side()
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.call('vqvae_memcpy_d2h', out.ctypes.data, self.ptr, self.nbytes, stream())
        return out

    def copy(self):
        out = DeviceArray(self.shape, self.dtype)
        _lib.call('vqvae_memcpy_d2d', out.ptr, self.ptr, self.nbytes, stream())
        return out

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError('size mismatch in copy_from')
        _lib.call('vqvae_memcpy_d2d', self.ptr, other.ptr, self.nbytes, stream())
        self.amax = getattr(other, 'amax', None)
        self._block.wver = _tick()
        return self

    def fill_zero(self):
        _lib.call('vqvae_memset', self.ptr, 0, self.nbytes, stream())
        self.amax = None
        self._block.wver = _tick()
        return self

    def __float__(self):
        if self.size != 1:
            raise TypeError('only size-1 arrays convert to float')
        return float(self.get().reshape(()))


def new_amax():
    """A zeroed group of absolute-maximum words for a launch's epilogue to raise (vqvae_conv1d_amax.out, ...)."""
    return zeros((_lib.AMAX_SLOTS,), np.uint32)


def absmax(x):
    """max |x| of a float32 device array as the device words matmul mode 'float32x2' passes around (one pass over x);
    remembered on the array."""
    if getattr(x, 'amax', None) is None:
        a = DeviceArray((_lib.AMAX_SLOTS,), np.uint32)
        _lib.call('vqvae_absmax', x.ptr, x.size, a.ptr, stream())
        x.amax = a
    return x.amax


def to_device(host, dtype=None):
    host = np.asarray(host)
    if dtype is None:
        dtype = host.dtype
    return DeviceArray(host.shape, dtype).set(host)


def copy_many(pairs):
    """dst.copy_from(src) for many (dst, src) pairs of float32 DeviceArrays in a handful of launches (vqvae_copy_list:
    64 copies per launch) -- the arena adoption of a model's ~340 parameters was a copy dispatch per parameter and arena."""
    pairs = [(d, s) for d, s in pairs if d.size]
    for d, s in pairs:
        if d.size != s.size or d.dtype != np.float32 or s.dtype != np.float32:
            raise ValueError('copy_many: float32 arrays of equal size')
    n = len(pairs)
    if not n:
        return
    dst = (C.c_void_p * n)(*[d.ptr for d, _ in pairs])
    src = (C.c_void_p * n)(*[s.ptr for _, s in pairs])
    cnt = (C.c_size_t * n)(*[d.size for d, _ in pairs])
    _lib.call('vqvae_copy_list', n, dst, src, cnt, stream())


def to_device_many(hosts):
    """Uploads a list of host float32 arrays in ONE transfer: they are laid end to end (each start 256-byte aligned) in
    one device buffer and returned as views of it."""
    hosts = [np.ascontiguousarray(h, np.float32) for h in hosts]
    offs, off = [], 0
    for h in hosts:
        offs.append(off)
        off += (h.size + 63) & ~63
    stage = np.zeros(max(off, 1), np.float32)
    for h, o in zip(hosts, offs):
        stage[o:o + h.size] = h.ravel()
    big = to_device(stage, np.float32)
    return [big.flat_view(o, h.size, h.shape) for h, o in zip(hosts, offs)]


def empty(shape, dtype=np.float32):
    return DeviceArray(shape, dtype)


def zeros(shape, dtype=np.float32):
    return DeviceArray(shape, dtype).fill_zero()


def is_device(x):
    return isinstance(x, DeviceArray)


def require_device(*arrays):
    """The NumPy/device mix guard of utils.py:183-186, plus: NumPy inputs are not
    computable here at all (no CPU path)."""
    for a in arrays:
        if not isinstance(a, DeviceArray):
            if isinstance(a, np.ndarray):
                raise ValueError(
                    'numpy and device arrays must not be used together / host arrays cannot '
                    'be computed on: call to_gpu() first (type: %s)' % type(a))
            raise TypeError('expected a DeviceArray, got %s' % type(a))


# --------------------------------------------------------------------------- #
# one growing scratch buffer shared by all entry points (single stream => the
# kernels that use it are ordered)
# --------------------------------------------------------------------------- #
def workspace(nbytes, slot='main'):
    """Growing scratch buffer; one per stream ('main', 'side').  A buffer is only ever
    replaced between uses on its own stream (callers size the side buffer before they start
    enqueueing work that uses it)."""
    ws = _state['ws'].get(slot)
    if ws is None or ws.nbytes < nbytes:
        _state['ws'][slot] = None
        ws = DeviceArray((int(nbytes * 1.25) // 4 + 64,), np.float32)
        _state['ws'][slot] = ws
    return ws
