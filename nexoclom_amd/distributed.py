"""Multi-GPU plumbing: one process per GPU, packets sharded by index, one sum of the image pair.

The reference has no parallelism; its natural data-parallel axis is the
packet index: it already splits a run into independent ``Output`` chunks
(initial_state/Input.py:243-246) and the only cross-chunk operation is the per-output-file image
sum of ModelImage.__init__ (data_simulation/ModelImage.py:96-98).  That sum is the one collective
here: ``ncclAllReduce`` on the device images, issued from libnexoclom_hip.so
(hip_api.Context.image_allreduce).

Partition: the run's packets form ONE global index space [0, N).  It is
cut into fixed chunks that do not depend on the number of ranks (``chunk_plan``); the host sampler
draws chunk k from the seeded generator ``seed + k`` (chunk 0 is the reference's own stream), the
device sampler is counter-based on the global index.  Rank r owns the contiguous index range
``shard_range(N, r, world)`` and takes exactly those rows of the chunks it overlaps, so 1 GPU, 2
GPUs and 8 GPUs integrate the same N packets and produce the same packet-count image.

``ControlPlane`` is the CPU-side side channel: a few dozen bytes per call over loopback TCP
(rendezvous of the RCCL unique id, agreement flags, scalar reductions).  No packet or pixel data
goes through it, except in ``allreduce_images_host``, which exists for the CPU tests and for
diagnostics and is never chosen silently: ``sharded_image`` raises when RCCL cannot be brought up.
Nothing here imports torch; ``python -m torch.distributed.run`` is only the process launcher whose
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment is read.
"""
import os
import secrets
import socket
import struct
import select
import tempfile
import threading
import time

import numpy as np

_MAGIC = b'NXCP1'
_OPS = {'SUM': np.add, 'MAX': np.maximum, 'MIN': np.minimum}


def shard_range(n, rank, world):
    """Contiguous index range [lo, hi) of rank's packets: sizes differ by at most one."""
    base, extra = divmod(int(n), int(world))
    lo = rank*base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_plan(total, chunk, lo=0, hi=None):
    """The pieces of the global chunk grid that the index range [lo, hi) covers.

    Chunk k spans [k*chunk, min((k+1)*chunk, total)) whatever the number of ranks.  Yields
    ``(k, chunk_start, chunk_len, a, b)``: rows [a, b) (global indices) of chunk k belong to the
    range.  With lo = 0 and hi = total this is the single-GPU chunk loop (Input.py:243-246)."""
    total, chunk = int(total), int(chunk)
    hi = total if hi is None else int(hi)
    if chunk < 1:
        raise ValueError('chunk must be positive')
    k = lo // chunk
    while k*chunk < hi:
        c0 = k*chunk
        clen = min(chunk, total - c0)
        a, b = max(lo, c0), min(hi, c0 + clen)
        if b > a:
            yield k, c0, clen, a, b
        k += 1


# ---- control plane ---------------------------------------------------------------------------
def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError('control-plane peer closed the connection')
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > (1 << 31):
        raise ConnectionError('control-plane message of absurd length')
    return _recv_exact(sock, n)


class ControlPlane:
    """Barrier / small reductions / byte broadcast across the ranks of one node.

    Star topology over loopback TCP: rank 0 listens on an ephemeral port and publishes
    ``port token`` in a rendezvous file named after MASTER_ADDR/MASTER_PORT (the launcher's own
    store owns that port, so it is used only as a job identifier); the other ranks poll the file,
    connect and prove the token.  Every method is a collective: all ranks must call it, in the
    same order.  Trivial when world == 1.

    Failure detection: every rank keeps a SECOND connection to rank 0 that carries nothing but
    goodbyes.  ``watch(callback)`` starts a daemon thread on it: a rank that fails calls
    ``announce_failure`` (or simply dies -- the kernel closes its socket), rank 0 tells everybody,
    and every rank's ``callback(reason)`` runs within milliseconds -- ``sharded`` code passes
    ``ctx.comm_request_abort`` so that a rank waiting inside an RCCL collective leaves it at once
    instead of at the collective deadline (hip_api.Context.comm_set_timeout).  ``close()`` says an
    orderly goodbye first, so a rank that merely finishes early raises no alarm.
    """

    def __init__(self, world=None, rank=None, timeout=300.0, rendezvous=None):
        env = os.environ
        self.world = int(env.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.rank = int(env.get('RANK', '0')) if rank is None else int(rank)
        self.local_rank = int(env.get('LOCAL_RANK', str(self.rank)))
        self.timeout = float(timeout)
        self._peers = {}           # rank 0: rank -> socket
        self._root = None          # other ranks: socket to rank 0
        self._watch_peers = {}     # rank 0: rank -> failure-channel socket
        self._watch_root = None    # other ranks: failure-channel socket to rank 0
        self._watcher = None
        self._lock = threading.Lock()
        self.failure = None        # the reason, once a peer's failure is known
        self._file = None
        if not 0 <= self.rank < self.world:
            raise ValueError(f'rank {self.rank} outside world of {self.world}')
        if self.world > 1:
            if rendezvous is None:
                job = '{}_{}_{}'.format(env.get('MASTER_ADDR', '127.0.0.1'),
                                        env.get('MASTER_PORT', '0'),
                                        env.get('TORCHELASTIC_RUN_ID', 'none'))
                job = ''.join(ch if ch.isalnum() or ch in '._-' else '_' for ch in job)
                rendezvous = os.path.join(tempfile.gettempdir(), f'nexoclom_cp_{job}.addr')
            self._file = rendezvous
            if self.rank == 0:
                self._serve()
            else:
                self._join()

    # -- rendezvous -------------------------------------------------------------------------
    def _serve(self):
        token = secrets.token_hex(16)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(('127.0.0.1', 0))
        srv.listen(self.world)
        srv.settimeout(self.timeout)
        tmp = f'{self._file}.{os.getpid()}'
        with open(tmp, 'w') as f:
            f.write(f'{srv.getsockname()[1]} {token}\n')
        os.replace(tmp, self._file)               # atomic: readers see old or new, never half
        try:
            while len(self._peers) + len(self._watch_peers) < 2*(self.world - 1):
                conn, _ = srv.accept()
                conn.settimeout(5.0)             # a stranger on the port must not stall the job
                try:
                    hello = _recv(conn)
                    magic, tok = hello[:5], hello[5:37].decode()
                    r, channel = struct.unpack('<II', hello[37:45])
                    table = (self._peers, self._watch_peers)[channel]
                    if magic != _MAGIC or tok != token or not 0 < r < self.world:
                        raise ConnectionError('bad hello')
                    if r in table:                # the rank lost a half-made pair and dials again
                        table.pop(r).close()
                except (ConnectionError, OSError, struct.error, UnicodeDecodeError, MemoryError,
                        IndexError):
                    conn.close()
                    continue
                conn.settimeout(self.timeout if channel == 0 else None)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _send(conn, _MAGIC + token.encode())
                table[r] = conn
        except socket.timeout:
            arrived = len(set(self._peers) & set(self._watch_peers)) + 1
            raise TimeoutError(f'control plane: only {arrived} of {self.world} '
                               f'ranks arrived within {self.timeout:.0f} s') from None
        finally:
            srv.close()
            try:
                os.remove(self._file)
            except OSError:
                pass

    def _join(self):
        deadline = time.monotonic() + self.timeout
        while True:
            try:
                with open(self._file) as f:
                    port, token = f.read().split()
                socks = []
                for channel in (0, 1):
                    sock = socket.create_connection(('127.0.0.1', int(port)), timeout=5.0)
                    socks.append(sock)
                    sock.settimeout(self.timeout)
                    _send(sock, _MAGIC + token.encode() + struct.pack('<II', self.rank, channel))
                    if _recv(sock) != _MAGIC + token.encode():
                        raise ConnectionError('bad reply')
                    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._root, self._watch_root = socks
                self._watch_root.settimeout(None)
                return
            except (OSError, ValueError, ConnectionError):
                for sock in locals().get('socks', ()):
                    sock.close()
                # no file yet, a stale file of an earlier job, or a foreign listener: look again
                if time.monotonic() > deadline:
                    raise TimeoutError(f'control plane: rank {self.rank} found no rank 0 via '
                                       f'{self._file} within {self.timeout:.0f} s') from None
                time.sleep(0.05)

    # -- collectives ------------------------------------------------------------------------
    def _exchange(self, payload, combine):
        """Every rank contributes ``payload``; rank 0 applies ``combine(list_by_rank) -> bytes``
        and everyone receives the result."""
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recv(self._peers[r]) for r in range(1, self.world)]
            out = combine(parts)
            for r in range(1, self.world):
                _send(self._peers[r], out)
            return out
        _send(self._root, payload)
        return _recv(self._root)

    def allreduce(self, values, op='SUM'):
        """Element-wise SUM / MAX / MIN of a float64 array over the ranks (in rank order, so
        every rank gets bit-identical results)."""
        fn = _OPS[op]
        arr = np.ascontiguousarray(values, dtype=np.float64)

        def combine(parts):
            acc = np.frombuffer(parts[0], dtype=np.float64).copy()
            for p in parts[1:]:
                acc = fn(acc, np.frombuffer(p, dtype=np.float64))
            return acc.tobytes()
        return np.frombuffer(self._exchange(arr.tobytes(), combine),
                             dtype=np.float64).reshape(arr.shape).copy()

    def reduce(self, value, op='SUM'):
        return float(self.allreduce([float(value)], op)[0])

    def barrier(self):
        self._exchange(b'', lambda parts: b'')

    def bcast_bytes(self, payload, n=None):
        """Rank 0's ``payload`` on every rank (``n``, the expected length, is checked)."""
        out = self._exchange(payload if self.rank == 0 else b'', lambda parts: parts[0])
        if n is not None and len(out) != n:
            raise ValueError(f'broadcast of {len(out)} bytes, expected {n}')
        return out

    def allgather_bytes(self, payload):
        """List, by rank, of every rank's ``payload``."""
        def combine(parts):
            return struct.pack('<I', len(parts)) + b''.join(
                struct.pack('<Q', len(p)) + p for p in parts)
        blob = self._exchange(bytes(payload), combine)
        (count,) = struct.unpack('<I', blob[:4])
        out, pos = [], 4
        for _ in range(count):
            (ln,) = struct.unpack('<Q', blob[pos:pos+8])
            out.append(blob[pos+8:pos+8+ln])
            pos += 8 + ln
        return out

    def allreduce_images_host(self, image, counts):
        """Sum an (image fp64, counts uint64) pair over ranks on the HOST.  For the CPU tests and
        diagnostics only -- the production path is the RCCL all-reduce on the device buffers and
        nothing falls back to this on its own."""
        image = np.asarray(image, dtype=np.float64)
        counts = np.asarray(counts)
        both = self.allreduce(np.concatenate([image.ravel(), counts.astype(np.float64).ravel()]))
        return (both[:image.size].reshape(image.shape),
                both[image.size:].astype(np.uint64).reshape(counts.shape))

    # -- RCCL bring-up ----------------------------------------------------------------------
    def init_rccl(self, ctx):
        """Create the RCCL communicator on a hip_api.Context: rank 0's unique id is broadcast
        over the control plane.  Raises hip_api.HipError on EVERY rank when any rank fails (two
        ranks on one device, librccl missing, ncclCommInitRank error): a collective that cannot
        run must stop the job, not change what is measured."""
        from . import hip_api
        me = f'{socket.gethostname()}|{ctx.bus_id()}'.encode()
        seats = [s.decode() for s in self.allgather_bytes(me)]
        for r, seat in enumerate(seats):
            if seats.index(seat) != r:
                raise hip_api.HipError(
                    f'ranks {seats.index(seat)} and {r} both sit on device {seat}: RCCL needs one '
                    f'process per GPU (WORLD_SIZE={self.world} but fewer devices are visible)')
        def agree(err):
            # same verdict on every rank: the first failing rank's message, or None
            problems = [p.decode() for p in self.allgather_bytes(err.encode())]
            bad = [r for r, p in enumerate(problems) if p]
            return f'rank {bad[0]}: {problems[bad[0]]}' if bad else None

        err, uid = '', bytes(hip_api.NXC_UNIQUE_ID_BYTES)
        if self.rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except hip_api.HipError as exc:
                err = str(exc)
        uid = self.bcast_bytes(uid, hip_api.NXC_UNIQUE_ID_BYTES)
        verdict = agree(err)
        if verdict is None:
            try:
                ctx.comm_init(uid, self.rank, self.world)
            except hip_api.HipError as exc:
                err = str(exc)
            verdict = agree(err)
        if verdict is not None:
            try:
                ctx.comm_destroy()
            except hip_api.HipError:
                pass
            raise hip_api.HipError(f'RCCL communicator could not be created ({verdict})')

    # -- failure detection --------------------------------------------------------------------
    def watch(self, callback):
        """Start the failure watcher: ``callback(reason)`` runs (once, on a daemon thread) as soon
        as any rank announces a failure or disappears without a goodbye."""
        if self.world == 1 or self._watcher is not None:
            return
        self._watcher = threading.Thread(target=self._watch_loop, args=(callback,), daemon=True,
                                         name='nexoclom-failure-watch')
        self._watcher.start()

    def _alarm(self, reason, callback):
        with self._lock:
            first, self.failure = self.failure is None, self.failure or reason
        if first and callback is not None:
            try:
                callback(self.failure)
            except Exception:                      # a watcher must not die of its callback
                pass

    def _watch_loop(self, callback):
        if self.rank != 0:
            sock = self._watch_root
            try:
                msg = sock.recv(256)
            except OSError:
                msg = b''
            if msg[:1] == b'B' or self._closing:
                return                              # rank 0 finished in good order
            self._alarm(msg[1:].decode(errors='replace') if msg[:1] == b'X'
                        else 'rank 0 is gone', callback)
            return
        live = dict(self._watch_peers)
        while live and not self._closing:
            try:
                ready, _, _ = select.select(list(live.values()), [], [], 0.25)
            except (OSError, ValueError):
                return
            for sock in ready:
                r = next(k for k, v in live.items() if v is sock)
                try:
                    msg = sock.recv(256)
                except OSError:
                    msg = b''
                del live[r]
                if msg[:1] == b'B':
                    continue                        # rank r finished in good order
                reason = (msg[1:].decode(errors='replace') if msg[:1] == b'X'
                          else f'rank {r} is gone (its process ended without a goodbye)')
                self._tell_all(b'X' + reason.encode()[:200])
                self._alarm(reason, callback)
                return

    def _tell_all(self, msg):
        for sock in self._watch_peers.values():
            try:
                sock.sendall(msg)
            except OSError:
                pass

    def announce_failure(self, reason):
        """This rank cannot go on: every other rank's watcher is told (through rank 0)."""
        if self.world == 1:
            return
        msg = b'X' + f'rank {self.rank}: {reason}'.encode()[:200]
        with self._lock:
            self.failure = self.failure or f'rank {self.rank}: {reason}'
        if self.rank == 0:
            self._tell_all(msg)
        elif self._watch_root is not None:
            try:
                self._watch_root.sendall(msg)
            except OSError:
                pass

    _closing = False

    def close(self):
        self._closing = True
        if self.failure is None:                   # an orderly goodbye raises no alarm
            if self.rank == 0:
                self._tell_all(b'B')
            elif self._watch_root is not None:
                try:
                    self._watch_root.sendall(b'B')
                except OSError:
                    pass
        socks = list(self._peers.values()) + list(self._watch_peers.values())
        socks += [s for s in (self._root, self._watch_root) if s is not None]
        for s in socks:
            try:
                s.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._root, self._watch_peers, self._watch_root = {}, None, {}, None


# ---- sharded run -----------------------------------------------------------------------------
def pick_device(cp, device=None):
    """The device index of this rank: LOCAL_RANK unless given.  A launcher that hands every
    process its own GPU through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (one visible device
    per process) is honoured: that device is index 0.  Ranks are never wrapped onto a shared
    device -- ``ControlPlane.init_rccl`` compares the PCI bus ids before RCCL is asked."""
    from . import hip_api
    ndev = hip_api.device_count()
    dev = cp.local_rank if device is None else int(device)
    masked = any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'))
    if device is None and ndev == 1 and masked:
        dev = 0
    if not 0 <= dev < ndev:
        raise hip_api.HipError(f'rank {cp.rank}: device {dev} does not exist ({ndev} visible); '
                               'launch one process per GPU')
    return dev


class guarded:
    """``with guarded(cp, ctx):`` around a sharded computation -- no rank can hang the others.
    The control plane's failure watcher ends this rank's wait on a collective as soon as a peer
    fails (``ctx.comm_request_abort``; past that, the collective deadline of
    hip_api.Context.comm_set_timeout holds); an exception raised in the block is announced to the
    peers before it propagates."""

    def __init__(self, cp, ctx):
        self.cp, self.ctx = cp, ctx

    def __enter__(self):
        if hasattr(self.cp, 'watch') and hasattr(self.ctx, 'comm_request_abort'):
            self.cp.watch(lambda reason: self.ctx.comm_request_abort())
        return self

    def __exit__(self, kind, exc, tb):
        if exc is not None and isinstance(exc, Exception):
            if hasattr(self.cp, 'announce_failure'):
                self.cp.announce_failure(f'{kind.__name__}: {exc}')
            if hasattr(self.ctx, 'comm_abort'):
                try:
                    self.ctx.comm_abort()
                except Exception:                   # noqa: BLE001 -- the first error is the news
                    pass
        return False


def merge_shards(img, cp, ctx, reduce='rccl'):
    """Sum the per-rank image pairs and the per-rank source totals into ``img`` on every rank
    (ModelImage.py:96-98 across GPUs).  reduce='rccl': ncclAllReduce of the device images
    (raises if the communicator cannot be created); reduce='host': sum over the control plane
    (CPU tests / diagnostics)."""
    if cp.world > 1:
        if reduce == 'rccl':
            cp.init_rccl(ctx)
            ctx.image_allreduce()
            image, counts = ctx.image_download()
            ctx.comm_destroy()
        elif reduce == 'host':
            image, counts = cp.allreduce_images_host(*ctx.image_download())
        else:
            raise ValueError("reduce must be 'rccl' or 'host'")
        img.image = np.array(image, dtype=np.float64)
        img.packet_image = counts.astype(float)
        totals = cp.allreduce([img.totalsource, img.npackets])
        img.totalsource = float(totals[0])
        img.npackets = int(totals[1])
    return img


def allreduce_small(values, cp, ctx, reduce='rccl'):
    """Sum over the ranks of a float64 array that lives on the host (an image pair, S radiances):
    reduce='rccl' -- ncclAllReduce through device scratch (nxc_allreduce_f64), with the
    communicator created for the call; reduce='host' -- over the control plane (CPU tests,
    diagnostics).  Integer-valued entries below 2^53 stay exact."""
    values = np.ascontiguousarray(values, dtype=np.float64)
    if cp.world == 1:
        return values
    if reduce == 'rccl':
        cp.init_rccl(ctx)
        out = ctx.allreduce(values)
        ctx.comm_destroy()
        return out
    if reduce == 'host':
        return cp.allreduce(values)
    raise ValueError("reduce must be 'rccl' or 'host'")


def merge_catalogue(img, cp, ctx, reduce='rccl'):
    """A ModelImage made from this rank's catalogue (finalize=False) becomes the image of the
    whole shared run: image, packet counts and source total summed over the ranks
    (ModelImage.py:96-98, where the reference sums over the files of one catalogue)."""
    n = img.image.size
    both = allreduce_small(np.concatenate([img.image.ravel(), img.packet_image.ravel(),
                                           [img.totalsource]]), cp, ctx, reduce)
    img.image = both[:n].reshape(img.image.shape).copy()
    img.packet_image = both[n:2*n].reshape(img.packet_image.shape).copy()
    img.totalsource = float(both[2*n])
    return img


def sharded_image(inputs, params, npackets, seed, cp=None, device=None, downcast=True,
                  sampler='device', packs_per_it=None, context=None, reduce='rccl'):
    """Multi-GPU ModelImage: every rank integrates and bins its contiguous shard of the global
    packet index range [0, npackets), then the image pairs are summed (one RCCL all-reduce).
    Packet i is the same packet whatever the number of ranks -- for both samplers, see the
    module docstring -- so the packet-count image is identical for 1, 2, 4 or 8 GPUs.

    Launch one process per GPU (e.g. ``python -m torch.distributed.run --nproc-per-node N``);
    returns a ModelImage holding the global image on every rank.  ``context``: an already
    created hip_api.Context (or, in the CPU tests, a stand-in with the same methods)."""
    from .ModelImage import ModelImage
    cp = cp or ControlPlane()
    if context is None:
        from . import hip_api
        context = hip_api.Context(pick_device(cp, device))
    with guarded(cp, context):
        if seed is None and sampler == 'device':
            # an unseeded run still is ONE run: every rank uses rank 0's fresh key
            from .Output import fresh_key
            seed = int.from_bytes(cp.bcast_bytes(fresh_key().to_bytes(8, 'little'), 8), 'little')
        lo, hi = shard_range(int(npackets), cp.rank, cp.world)
        img = ModelImage(inputs, params, npackets=int(npackets), shard=(lo, hi), seed=seed,
                         context=context, downcast=downcast, sampler=sampler,
                         packs_per_it=packs_per_it, finalize=False)
        merge_shards(img, cp, context, reduce)
        img.finalize()
    return img
