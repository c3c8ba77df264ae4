"""Communicators for sharded runs: one process per GPU, results gathered once at the end.

The path partitions by evaluation (toy datasets, scan points) and has no data-path collective; the ONE exchange is
the gather of the per-rank fp64 result vectors.  The reference has no counterpart (its scans
are Python loops, blueice/inference.py:49-50,424-432; its only parallelism farms template building out to
processes, blueice/parallel.py:47-103).

Three implementations of one small interface (`rank`, `world`, `all_gather`, `all_reduce`, `broadcast_bytes`,
`barrier`, `close`):

  SoloCommunicator     a single process.
  SocketCommunicator   TCP over the loopback interface, star-shaped through rank 0.  Host memory only: the
                       bootstrap channel (rendezvous, RCCL unique id), the CPU rehearsal of the sharded paths,
                       and the fall-back gather when RCCL cannot be initialised on a box.
  RcclCommunicator     RCCL bound directly with ctypes (librccl.so: ncclGetUniqueId / ncclCommInitRank /
                       ncclAllGather / ncclAllReduce), collectives enqueued on the DeviceContext's own HIP stream
                       (`bi_stream`) between device buffers -- the results of `bi_run_plan` go from HBM over xGMI
                       to every rank's HBM with no host hop and no second runtime in the process.

Launch: ranks are plain processes started before any of them touches the GPU -- `python -m blueice_amd.launch
--nproc N script.py ...`, or any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
(torch.distributed.run does).  Rendezvous: rank 0 listens on an ephemeral port and publishes it in a small file
whose name every rank derives from the same environment (BLUEICE_AMD_RDZV, or MASTER_PORT + parent pid).
"""
import ctypes as C
import os
import socket
import struct
import tempfile
import time

import numpy as np

__all__ = ['SoloCommunicator', 'SocketCommunicator', 'RcclCommunicator', 'connect', 'CommError', 'CommInitTimeout']

_OPS = ('sum', 'max', 'min', 'bor')


class CommError(RuntimeError):
    pass


class CommInitTimeout(CommError):
    """ncclCommInitRank did not return on some rank.  `stuck` is True on the ranks whose init thread is still inside
    RCCL: such a process cannot shut down cleanly and must leave with os._exit (bench.py does)."""
    stuck = False


def _reduce(parts, op):
    stack = np.stack(parts)
    if op == 'sum':
        out = stack[0].copy()
        for p in stack[1:]:               # fixed rank order: every rank computes the same bits
            out = out + p
        return out
    if op == 'max':
        return stack.max(axis=0)
    if op == 'min':
        return stack.min(axis=0)
    if op == 'bor':
        return np.bitwise_or.reduce(stack.astype(np.int64), axis=0).astype(stack.dtype)
    raise ValueError("op must be one of %s" % (_OPS,))


class SoloCommunicator:
    rank, world = 0, 1
    kind = 'solo'

    def all_gather(self, local):
        return np.asarray(local)[None, ...].copy()

    def all_reduce(self, local, op='sum'):
        return np.array(local, copy=True)

    def broadcast_bytes(self, data=None):
        return data

    def barrier(self):
        pass

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------------
# sockets
# ---------------------------------------------------------------------------------------------------
def _send(sock, payload):
    try:
        sock.sendall(struct.pack('<q', len(payload)) + payload)
    except OSError as e:                                   # socket.timeout included
        raise CommError("send to peer failed: %s" % (e or type(e).__name__))


def _recv_exact(sock, n):
    chunks, got = [], 0
    while got < n:
        try:
            b = sock.recv(min(n - got, 1 << 20))
        except OSError as e:                               # socket.timeout included: a peer that never answers
            raise CommError("no answer from peer: %s" % (e or type(e).__name__))
        if not b:
            raise CommError("peer closed the connection")
        chunks.append(b)
        got += len(b)
    return b''.join(chunks)


def _recv(sock):
    (n,) = struct.unpack('<q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def rendezvous_path():
    """The file rank 0 publishes its port in; the same for every rank of one launch."""
    explicit = os.environ.get('BLUEICE_AMD_RDZV')
    if explicit:
        return explicit
    key = '%s_%s_%s_%d' % (os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '0'),
                           os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())
    key = ''.join(ch if ch.isalnum() or ch in '._-' else '_' for ch in key)
    return os.path.join(tempfile.gettempdir(), 'blueice_amd_rdzv_%d_%s' % (os.getuid(), key))


class SocketCommunicator:
    """Host-side collectives over loopback TCP, star-shaped through rank 0.  Payloads here are a few MB at most
    (10^6 doubles at configs[3]); the point is correctness and zero dependencies, not bandwidth."""

    kind = 'socket'

    def __init__(self, rank, world, path=None, timeout=180.0, host='127.0.0.1', io_timeout=None):
        # timeout: the rendezvous; io_timeout: how long a receive waits for a peer afterwards (default: the same)
        io_timeout = timeout if io_timeout is None else io_timeout
        self.rank, self.world = int(rank), int(world)
        self._peers = []          # rank 0: socket of every other rank, by rank
        self._up = None           # other ranks: socket to rank 0
        self._path = path or rendezvous_path()
        deadline = time.monotonic() + timeout
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((host, 0))
            srv.listen(self.world)
            port = srv.getsockname()[1]
            tmp = '%s.%d.tmp' % (self._path, os.getpid())
            with open(tmp, 'w') as f:
                f.write('%s %d %d\n' % (host, port, os.getpid()))
            os.replace(tmp, self._path)                        # atomic: readers see all of it or the previous file
            peers = {}
            srv.settimeout(1.0)
            while len(peers) < self.world - 1:
                if time.monotonic() > deadline:
                    raise CommError("rendezvous: %d of %d ranks connected within %.0f s" % (len(peers) + 1, self.world, timeout))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                r, w = struct.unpack('<ii', _recv(conn))
                if w != self.world or not 0 < r < self.world or r in peers:
                    conn.close()
                    raise CommError("rendezvous: unexpected peer (rank %d of %d)" % (r, w))
                conn.settimeout(io_timeout)
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
            try:
                os.unlink(self._path)
            except OSError:
                pass
            for p in self._peers:
                _send(p, b'go')
        else:
            last = None
            while True:
                if time.monotonic() > deadline:
                    raise CommError("rendezvous: rank %d could not reach rank 0 via %s (%s)" % (self.rank, self._path, last))
                try:
                    with open(self._path) as f:
                        h, port, _ = f.read().split()
                    s = socket.create_connection((h, int(port)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    _send(s, struct.pack('<ii', self.rank, self.world))
                    if _recv(s) != b'go':
                        raise CommError("bad handshake")
                    s.settimeout(io_timeout)
                    self._up = s
                    break
                except (OSError, ValueError, CommError) as e:     # no file yet, a stale file, or rank 0 not listening yet
                    last = e
                    time.sleep(0.05)

    # -- primitives ---------------------------------------------------------------------------
    def _gather_bytes(self, payload):
        """rank 0 -> list of every rank's payload (by rank); other ranks -> None."""
        if self.rank == 0:
            return [payload] + [_recv(p) for p in self._peers]
        _send(self._up, payload)
        return None

    def broadcast_bytes(self, data=None):
        if self.world == 1:
            return data
        if self.rank == 0:
            for p in self._peers:
                _send(p, data)
            return data
        return _recv(self._up)

    def all_gather(self, local):
        """Equal-shaped arrays from every rank -> [world, ...] on every rank."""
        local = np.ascontiguousarray(local)
        if self.world == 1:
            return local[None, ...].copy()
        parts = self._gather_bytes(local.tobytes())
        blob = self.broadcast_bytes(b''.join(parts) if self.rank == 0 else None)
        return np.frombuffer(blob, dtype=local.dtype).reshape((self.world,) + local.shape).copy()

    def all_reduce(self, local, op='sum'):
        local = np.ascontiguousarray(local)
        if self.world == 1:
            return local.copy()
        parts = self._gather_bytes(local.tobytes())
        if self.rank == 0:
            out = _reduce([np.frombuffer(p, dtype=local.dtype).reshape(local.shape) for p in parts], op)
            self.broadcast_bytes(np.ascontiguousarray(out, dtype=local.dtype).tobytes())
            return out.astype(local.dtype, copy=False)
        return np.frombuffer(self.broadcast_bytes(), dtype=local.dtype).reshape(local.shape).copy()

    def barrier(self):
        self.all_reduce(np.zeros(1))

    def close(self):
        for s in self._peers + ([self._up] if self._up is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = [], None


# ---------------------------------------------------------------------------------------------------
# RCCL, bound directly
# ---------------------------------------------------------------------------------------------------
class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_ubyte * 128)]          # NCCL_UNIQUE_ID_BYTES (rccl.h); raw bytes, NULs included


_NCCL_INT64, _NCCL_FLOAT64 = 4, 8                        # ncclDataType_t (rccl.h)
_NCCL_OPS = {'sum': 0, 'max': 2, 'min': 3}               # ncclRedOp_t
_rccl = None


def load_rccl():
    """librccl.so with prototypes set; CommError if it cannot be loaded."""
    global _rccl
    if _rccl is not None:
        return _rccl
    last = None
    for name in (os.environ.get('BLUEICE_AMD_RCCL'), 'librccl.so.1', '/opt/rocm/lib/librccl.so.1', 'librccl.so'):
        if not name:
            continue
        try:
            lib = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError as e:
            last = e
    else:
        raise CommError("cannot load librccl.so: %s" % last)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [i]
    for name, args in (('ncclGetVersion', [C.POINTER(i)]), ('ncclGetUniqueId', [C.POINTER(_UniqueId)]),
                       ('ncclCommInitRank', [C.POINTER(vp), i, _UniqueId, i]), ('ncclCommDestroy', [vp]),
                       ('ncclAllGather', [vp, vp, sz, i, vp, vp]), ('ncclAllReduce', [vp, vp, sz, i, i, vp, vp])):
        fn = getattr(lib, name)
        fn.restype = i
        fn.argtypes = args
    # what the communicator says about itself once it exists (rccl.h): bound when the library has them
    for name in ('ncclCommCount', 'ncclCommCuDevice', 'ncclCommUserRank'):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = i
            fn.argtypes = [vp, C.POINTER(i)]
    _rccl = lib
    return lib


class RcclCommunicator:
    """Device-side collectives on the context's stream.  The constructor only calls ncclCommInitRank with a unique id the
    caller has already distributed (`exchange_unique_id`): it touches no socket, so it can run on a helper thread while
    the bootstrap channel stays with the main thread.  `boot` is the connected host communicator (host-side odds and ends
    such as the max-over-ranks of a wall time); attach it after construction."""

    kind = 'rccl'

    def __init__(self, ctx, rank, world, unique_id, boot=None):
        self.ctx, self.boot = ctx, boot
        self.rank, self.world = int(rank), int(world)
        self._lib = load_rccl()
        self._comm = C.c_void_p()
        self._send = self._recv = None
        if len(unique_id) != 128:
            raise CommError("unique id of %d bytes, expected 128" % len(unique_id))
        uid = _UniqueId()
        C.memmove(C.addressof(uid), unique_id, 128)
        scratch = ctx.device_alloc(16)                 # also makes the context's GPU the calling thread's current device
        scratch.free()
        self._ok(self._lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), 'ncclCommInitRank')
        v = C.c_int()
        self._lib.ncclGetVersion(C.byref(v))
        self.version = v.value
        # The communicator's own account of itself: how many ranks it spans, which one this is, which GPU it sits on.  A run
        # reports these (bench.py: config.rccl_ranks / rank_devices), so that a log answers "did RCCL see N ranks, one per
        # GPU" without anybody having watched the run; a count that is not the world size is an error here and now.
        self.n_ranks, self.user_rank, self.device = self._query('ncclCommCount'), self._query('ncclCommUserRank'), self._query('ncclCommCuDevice')
        if self.n_ranks is not None and self.n_ranks != self.world:
            raise CommError("ncclCommCount says %d ranks, the launcher started %d" % (self.n_ranks, self.world))
        if self.user_rank is not None and self.user_rank != self.rank:
            raise CommError("ncclCommUserRank says rank %d, the launcher says %d" % (self.user_rank, self.rank))

    def _query(self, name):
        fn = getattr(self._lib, name, None)
        if fn is None:
            return None
        v = C.c_int(-1)
        self._ok(fn(self._comm, C.byref(v)), name)
        return int(v.value)

    def _ok(self, rc, what):
        if rc != 0:
            raise CommError("%s failed: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    # -- device form: pointers in, nothing leaves HBM -------------------------------------------
    def all_gather_device(self, send_ptr, recv_ptr, count, dtype=np.float64):
        """recv[r * count : (r + 1) * count] = rank r's send[0:count], enqueued on the context stream."""
        code = _NCCL_FLOAT64 if np.dtype(dtype) == np.float64 else _NCCL_INT64
        self._ok(self._lib.ncclAllGather(C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count), code, self._comm,
                                         C.c_void_p(self.ctx.stream)), 'ncclAllGather')

    def all_reduce_device(self, send_ptr, recv_ptr, count, op='sum', dtype=np.float64):
        code = _NCCL_FLOAT64 if np.dtype(dtype) == np.float64 else _NCCL_INT64
        self._ok(self._lib.ncclAllReduce(C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count), code, _NCCL_OPS[op],
                                         self._comm, C.c_void_p(self.ctx.stream)), 'ncclAllReduce')

    # -- host form (same interface as SocketCommunicator): staged through two device buffers ------
    def _staging(self, nbytes):
        if self._send is None or self._send.nbytes < nbytes:
            for b in (self._send, self._recv):
                if b is not None:
                    b.free()
            self._send = self.ctx.device_alloc(nbytes)
            self._recv = self.ctx.device_alloc(nbytes * self.world)
        return self._send, self._recv

    def all_gather(self, local):
        local = np.ascontiguousarray(local)
        if local.dtype not in (np.float64, np.int64):
            raise TypeError("RCCL path carries float64 / int64")
        send, recv = self._staging(max(local.nbytes, 16))
        send.from_host(local)
        self.all_gather_device(send.ptr, recv.ptr, local.size, local.dtype)
        return recv.to_host(local.dtype, local.size * self.world).reshape((self.world,) + local.shape)

    def all_reduce(self, local, op='sum'):
        local = np.ascontiguousarray(local)
        if op == 'bor':                                        # no bitwise reduction in RCCL's fp path: gather, then OR
            return np.bitwise_or.reduce(self.all_gather(local.astype(np.int64)), axis=0).astype(local.dtype)
        if local.dtype not in (np.float64, np.int64):
            raise TypeError("RCCL path carries float64 / int64")
        send, recv = self._staging(max(local.nbytes, 16))
        send.from_host(local)
        self.all_reduce_device(send.ptr, recv.ptr, local.size, op, local.dtype)
        return recv.to_host(local.dtype, local.size).reshape(local.shape)

    def broadcast_bytes(self, data=None):
        return self.boot.broadcast_bytes(data) if self.boot is not None else data

    def barrier(self):
        self.ctx.sync()
        self.all_reduce(np.zeros(1))

    def close(self):
        for b in (self._send, self._recv):
            if b is not None:
                b.free()
        self._send = self._recv = None
        if self._comm:
            self.ctx.sync()
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
        if self.boot is not None:
            self.boot.close()


def describe(comm, requested='rccl', local_device=None):
    """What a run should say about its gather, agreed over all ranks: the kind that was requested and the kind that runs,
    the number of ranks RCCL itself reports (ncclCommCount; None without RCCL), the GPU of every rank as RCCL sees it
    (ncclCommCuDevice; else the launcher's LOCAL_RANK / `local_device`), and why a fallback was taken.  Collective: every
    rank calls it."""
    kind = getattr(comm, 'kind', 'none')
    host = comm.boot if (kind == 'rccl' and getattr(comm, 'boot', None) is not None) else comm
    dev = getattr(comm, 'device', None)
    if dev is None:
        dev = int(os.environ.get('LOCAL_RANK', -1)) if local_device is None else int(local_device)
    devices = [int(d) for d in np.ravel(host.all_gather(np.array([float(dev)])))] if host is not None and hasattr(host, 'all_gather') else [dev]
    return dict(backend_requested=requested, gather_kind=kind, rccl_ranks=getattr(comm, 'n_ranks', None) if kind == 'rccl' else None,
                rccl_version=getattr(comm, 'version', None) if kind == 'rccl' else None, rank_devices=devices,
                rank_devices_source='ncclCommCuDevice' if kind == 'rccl' and getattr(comm, 'device', None) is not None else 'launcher',
                gather_fallback_reason=getattr(comm, 'fallback_reason', '') or None)


def exchange_unique_id(boot, lib):
    """Rank 0 draws the RCCL unique id and EVERY rank receives either it or the reason there is none: rank 0 always
    broadcasts -- one status byte, then the 128 id bytes or a message -- so no peer waits for an id that will never come
    (a rank 0 that failed in ncclGetUniqueId used to leave the others blocked on the channel).
    -> (id bytes or None, reason)."""
    payload = None
    if boot.rank == 0:
        try:
            uid = _UniqueId()
            rc = lib.ncclGetUniqueId(C.byref(uid))
            if rc == 0:
                payload = b'\x01' + C.string_at(C.addressof(uid), 128)
            else:
                payload = b'\x00' + ('ncclGetUniqueId failed: %s' % lib.ncclGetErrorString(rc).decode()).encode()
        except Exception as e:                                # ctypes / loader trouble: still tell the others
            payload = b'\x00' + ('ncclGetUniqueId raised %r' % (e,)).encode()
    blob = boot.broadcast_bytes(payload)
    if blob[:1] == b'\x01' and len(blob) == 129:
        return blob[1:], ''
    return None, (blob[1:].decode(errors='replace') if blob[:1] == b'\x00' else 'malformed unique-id message from rank 0')


_INIT_OK, _INIT_FAILED, _INIT_STUCK = 0.0, 1.0, 2.0


def connect(ctx=None, backend='rccl', rank=None, world=None, timeout=180.0):
    """The communicator of this process, from the launcher's environment (RANK / WORLD_SIZE, default one process).

    backend 'rccl' needs the rank's DeviceContext.  If RCCL cannot be loaded, rank 0 cannot draw a unique id, or
    ncclCommInitRank RETURNS an error on any rank, every rank falls back to the socket communicator together
    (`.kind == 'socket'`, `.fallback_reason` says why) -- a gather of a few MB must not be what fails a run.  If
    ncclCommInitRank does not return within `timeout` on some rank, every rank raises CommInitTimeout: a process with
    a thread stuck inside RCCL must end (non-zero), not carry on beside it.  backend 'socket': host-side only."""
    rank = int(os.environ.get('RANK', 0)) if rank is None else int(rank)
    world = int(os.environ.get('WORLD_SIZE', 1)) if world is None else int(world)
    if world == 1 and backend != 'rccl':
        return SoloCommunicator()
    # a receive on the bootstrap channel may have to wait for a peer that sits out its own `timeout` first (the agreement
    # after ncclCommInitRank): its patience is clearly longer than that, so the slow rank's answer is never cut off
    boot = SocketCommunicator(rank, world, timeout=timeout, io_timeout=2.0 * timeout + 30.0) if world > 1 else SoloCommunicator()
    if backend == 'socket':
        return boot
    if backend != 'rccl':
        raise ValueError("backend must be 'rccl' or 'socket'")
    if ctx is None:
        raise ValueError("the rccl backend needs this rank's DeviceContext")

    def fall_back(reason):
        boot.fallback_reason = reason
        return boot

    reason, lib = '', None
    try:
        lib = load_rccl()
    except CommError as e:
        reason = str(e)
    # agree before anybody enters ncclCommInitRank (which blocks until every rank has arrived)
    if boot.all_reduce(np.array([1.0 if reason else 0.0]), 'max')[0] > 0:
        return fall_back(reason or 'librccl.so could not be loaded on another rank')
    unique_id, reason = exchange_unique_id(boot, lib)             # on the main thread: the channel has one user
    if unique_id is None:
        return fall_back(reason)
    # ncclCommInitRank blocks until every rank has arrived; it runs on a helper thread that touches no socket, so the
    # main thread can give up on it after `timeout` and still talk to the other ranks
    import threading
    box = {}

    def build():
        try:
            box['comm'] = RcclCommunicator(ctx, rank, world, unique_id)
        except BaseException as e:                         # CommError, OSError from ctypes, ...
            box['error'] = e

    th = threading.Thread(target=build, name='rccl-init', daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        state, reason = _INIT_STUCK, 'ncclCommInitRank did not return within %.0f s on rank %d' % (timeout, rank)
    elif 'error' in box:
        state, reason = _INIT_FAILED, str(box['error'])
    else:
        state = _INIT_OK
    worst = boot.all_reduce(np.array([state]), 'max')[0]
    if worst >= _INIT_STUCK:
        err = CommInitTimeout(reason or 'ncclCommInitRank did not return within %.0f s on another rank' % timeout)
        err.stuck = th.is_alive()
        try:
            boot.close()
        finally:
            raise err
    if worst >= _INIT_FAILED:
        # a communicator that exists on some ranks only is not usable and not safely destroyable: forget it
        return fall_back(reason or 'ncclCommInitRank failed on another rank')
    comm = box['comm']
    comm.boot = boot
    return comm
