"""Data-parallel communicators.

``RcclCommunicator`` -- one process per GPU, gradients exchanged by one in-place
RCCL all-reduce(sum) of the flat gradient arena over xGMI (C ABI
vqvae_comm_*).  Replaces Link.addgrads + Link.copyparams of the reference's
single-process multi-GPU updater (updaters.py:71-77).

Rendezvous (single node): rank 0 creates the ncclUniqueId and publishes it
through a file in a per-user 0700 directory, keyed by the job (VQVAE_RDZV_ID from bench.py's
own spawner, or MASTER_PORT + launcher pid under `python -m torch.distributed.run`).  Either
launcher only supplies RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* -- torch itself is never
imported in a GPU process (it bundles its own HIP runtime).

The world_size-2 CPU tests drive the same shard / sum / lr logic through a
gloo-backed communicator with this interface that lives in tests/dp_worker.py
(torch is test tooling only; nothing in this package imports it).
"""
import ctypes as C
import os
import sys
import time

import numpy as np


def shard(batch, rank, size):
    """The reference's strided split ``batch[i::n]`` (updaters.py:37-38)."""
    return batch[rank::size]


def scaled_alpha(lr, size):
    """train.py:101: Adam(params.lr / len(args.gpus)) -- gradients are SUMMED
    over replicas (updaters.py:72), the step size is divided instead."""
    return lr / size


class SingleCommunicator(object):
    rank, size = 0, 1

    def ranks_seen(self):
        return None          # no communicator, no RCCL: nothing to report

    def allreduce_grad(self, flat, stream=None):
        return flat

    def barrier(self):
        pass

    def max_scalar(self, v):
        return v

    def comm_time_ms(self):
        return 0.0, 0


def _rendezvous_dir():
    """Per-user 0700 directory for the id hand-off (not a world-writable /tmp name: another local
    user could otherwise pre-create or symlink the file the ranks trust)."""
    base = os.environ.get('XDG_RUNTIME_DIR') or '/tmp'
    d = os.path.join(base, 'vqvae_rccl_%d' % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError('rendezvous directory %s is not a private directory of this user' % d)
    return d


def _rendezvous_path():
    """One file per job: keyed by VQVAE_RDZV_ID when the launcher sets one (bench.py's own
    spawner does), else by what `python -m torch.distributed.run` provides."""
    key = os.environ.get('VQVAE_RDZV_ID')
    if not key:
        key = '%s_%s_%s_%d' % (os.environ.get('MASTER_PORT', '0'),
                               os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
                               os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), os.getppid())
    key = ''.join(c if (c.isalnum() or c in '-_') else '_' for c in key)
    return os.path.join(_rendezvous_dir(), 'uid_' + key)


def _publish(path, raw):
    """Atomic, exclusive creation (a stale file of the same name is replaced, never followed)."""
    tmp = '%s.%d.tmp' % (path, os.getpid())
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
    try:
        os.write(fd, raw)
    finally:
        os.close(fd)
    os.rename(tmp, path)


def _read_owned(path):
    fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    try:
        if os.fstat(fd).st_uid != os.getuid():
            raise RuntimeError('rendezvous file %s is not owned by this user' % path)
        return os.read(fd, 128)
    finally:
        os.close(fd)


def exchange_unique_id(rank, make_id, timeout=300.0):
    """The file rendezvous (no GPU, no library involved): rank 0 calls ``make_id()`` -> 128 bytes and publishes them,
    every other rank waits for the file and reads it.  Returns (path, the 128 bytes).  The caller (rank 0) removes the
    file once every rank has joined."""
    path = _rendezvous_path()
    if rank == 0:
        raw = make_id()
        _publish(path, raw)
        return path, raw
    t0 = time.time()

    def fresh():
        # a file left behind by a crashed earlier job with the same key is not ours: ranks of
        # one job start within seconds of each other, rank 0's file cannot be minutes old
        try:
            return os.path.getmtime(path) > t0 - 120.0
        except OSError:
            return False
    while not fresh():
        if time.time() - t0 > timeout:
            raise RuntimeError('RCCL rendezvous timed out waiting for %s' % path)
        time.sleep(0.05)
    return path, _read_owned(path)


def gpu_numa_node(pci_bus_id):
    """NUMA node of a PCI device from sysfs ('0000:c1:00.0' -> int, None when unknown / -1)."""
    try:
        with open('/sys/bus/pci/devices/%s/numa_node' % pci_bus_id.lower()) as fh:
            node = int(fh.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def bind_to_numa_node(node, allowed_cpus=None):
    """Prefer ``node`` for this process's future page allocations (set_mempolicy(MPOL_PREFERRED)) and run on its
    cores (``allowed_cpus``: the set to choose from, default the current affinity): the page-locked input buffers and
    the Python heap of a rank then live next to its GPU.  Returns a dict describing what was done (for the bench
    line); never raises."""
    out = {'numa_node': node, 'mempolicy': None, 'cpus': None}
    if node is None:
        return out
    try:
        with open('/sys/devices/system/node/node%d/cpulist' % node) as fh:
            cpus = set()
            for part in fh.read().strip().split(','):
                a, _, b = part.partition('-')
                cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & set(allowed_cpus if allowed_cpus is not None else os.sched_getaffinity(0))
        if allowed:
            os.sched_setaffinity(0, allowed)
            out['cpus'] = [min(allowed), max(allowed), len(allowed)]
    except (OSError, ValueError, AttributeError):
        pass
    try:
        libc = C.CDLL(None, use_errno=True)
        mask = (C.c_ulong * 16)()
        mask[node // (8 * C.sizeof(C.c_ulong))] = 1 << (node % (8 * C.sizeof(C.c_ulong)))
        rc = libc.syscall(238 if os.uname().machine == 'x86_64' else 237, 1, mask, C.c_ulong(16 * 8 * C.sizeof(C.c_ulong)))   # set_mempolicy(MPOL_PREFERRED)
        out['mempolicy'] = 'preferred' if rc == 0 else 'set_mempolicy errno %d' % C.get_errno()
    except Exception as e:              # no libc / seccomp: report, carry on
        out['mempolicy'] = 'unavailable (%s)' % type(e).__name__
    return out


class RcclCommunicator(object):
    always_reduce = True      # a 1-rank communicator still exercises the all-reduce (self-test)
    capture_safe = True       # allreduce_grad only enqueues ncclAllReduce on the given stream: nothing host-side to lose in a recorded step

    def __init__(self, rank=None, size=None, device=None, timeout=300.0):
        from . import _lib, backend
        self.rank = int(os.environ.get('RANK', 0)) if rank is None else rank
        self.size = int(os.environ.get('WORLD_SIZE', 1)) if size is None else size
        local = int(os.environ.get('LOCAL_RANK', self.rank)) if device is None else device
        backend.init(local)
        self._lib, self._backend = _lib, backend
        def make_id():
            buf = C.create_string_buffer(128)
            _lib.call('vqvae_comm_unique_id', buf)
            return buf.raw
        path, raw = exchange_unique_id(self.rank, make_id, timeout)
        idbuf = C.create_string_buffer(raw, 128)
        comm = C.c_void_p()
        # RCCL prints a version banner on first init; keep stdout clean for callers that
        # parse it (bench.py prints exactly one JSON line): route fd 1 to stderr meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            _lib.call('vqvae_comm_init', C.byref(comm), self.size, self.rank, idbuf)
        finally:
            try:
                C.CDLL(None).fflush(None)      # the banner sits in libc's stdout buffer: push it out NOW,
            except Exception:                  # while fd 1 still points at stderr
                pass
            os.dup2(saved, 1)
            os.close(saved)
        self._comm = comm
        self.time_comm = False
        self._timed = []
        self._scalar = backend.zeros((1,), np.float32)
        self.barrier()
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    def allreduce_grad(self, flat, stream=None):
        """In-place sum over ranks of a flat fp32 DeviceArray, enqueued on ``stream`` (default: the
        process stream).  With ``time_comm`` set, every call is bracketed by HIP events on its
        stream; comm_time_ms() adds them up (after a synchronize) -- bench.py's comm_ms_per_step."""
        st = self._backend.stream() if stream is None else stream
        if self.time_comm:
            e0 = self._backend.Event().record(st)
        self._lib.call('vqvae_comm_allreduce_sum_f32', self._comm, flat.ptr, flat.size, st)
        if self.time_comm:
            self._timed.append((e0, self._backend.Event().record(st)))
        return flat

    def comm_time_ms(self):
        """(total ms, calls) of the timed all-reduces so far; call after backend.synchronize()."""
        tot = 0.0
        ms = C.c_float(0)
        for e0, e1 in self._timed:
            self._lib.call('vqvae_event_elapsed_ms', C.byref(ms), e0.h, e1.h)
            tot += ms.value
        n = len(self._timed)
        self._timed = []
        return tot, n

    def barrier(self):
        self._scalar.fill_zero()
        self._lib.call('vqvae_comm_allreduce_sum_f32', self._comm, self._scalar.ptr, 1,
                       self._backend.stream())
        self._backend.synchronize()

    def max_scalar(self, v):
        self._scalar.set(np.array([v], np.float32))
        self._lib.call('vqvae_comm_allreduce_max_f32', self._comm, self._scalar.ptr, 1,
                       self._backend.stream())
        return float(self._scalar.get()[0])

    def ranks_seen(self):
        """(ncclCommCount, all-reduced sum of ones): how many ranks RCCL itself reports and how
        many actually contributed to a collective."""
        n = C.c_int(0)
        self._lib.call('vqvae_comm_count', self._comm, C.byref(n))
        self._scalar.set(np.array([1.0], np.float32))
        self._lib.call('vqvae_comm_allreduce_sum_f32', self._comm, self._scalar.ptr, 1,
                       self._backend.stream())
        return n.value, int(round(float(self._scalar.get()[0])))

    def self_test(self, arena=None):
        """Start-up check of the exchange itself, before any step depends on it: the all-reduced sum / max / min of
        the rank ids must be n(n-1)/2, n-1 and 0, and (with ``arena`` = the parameter arena) every rank must hold the
        same parameters.  Returns a dict for bench.py's multi_rank_diagnostics; raises when a collective lies."""
        n, r = self.size, self.rank
        from . import _lib
        self._scalar.set(np.array([float(r)], np.float32))
        self._lib.call('vqvae_comm_allreduce_sum_f32', self._comm, self._scalar.ptr, 1, self._backend.stream())
        s = float(self._scalar.get()[0])
        hi, lo = self.max_scalar(float(r)), -self.max_scalar(-float(r))
        out = {'rank_id_sum': s, 'rank_id_max': hi, 'rank_id_min': lo, 'expected_sum': n * (n - 1) / 2.0}
        if s != n * (n - 1) / 2.0 or hi != n - 1 or lo != 0:
            raise RuntimeError('RCCL self-test failed on rank %d: sum / max / min of the rank ids = %r / %r / %r with %d ranks'
                               % (r, s, hi, lo, n))
        if arena is not None:
            tot = self._backend.zeros((1,), np.float32)
            ws = self._backend.workspace(4096 * 4)
            _lib.call('vqvae_sum', arena.ptr, arena.size, 1.0, tot.ptr, ws.ptr, ws.nbytes, self._backend.stream())
            v = float(tot.get()[0])
            chi, clo = self.max_scalar(v), -self.max_scalar(-v)
            out.update(param_checksum_max=chi, param_checksum_min=clo)
            if chi != clo:
                raise RuntimeError('data-parallel replicas start from different parameters: checksum %r .. %r across ranks'
                                   % (clo, chi))
        return out

    def close(self):
        if self._comm is not None:
            self._lib.call('vqvae_comm_destroy', self._comm)
            self._comm = None
