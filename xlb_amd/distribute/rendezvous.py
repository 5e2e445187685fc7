"""Control plane of a multi-process (one process per GPU) run over plain TCP sockets: rendezvous,
unique-id broadcast, barriers, small reductions and result gathering.  Standard library only.

The reference relies on ``jax.distributed`` / a single multi-device process for this
(xlb/distribute/distribute.py:82-105); here every rank is its own process, started by
``python -m torch.distributed.run`` or by ``bench.py``'s own launcher, and described by the
usual environment: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT.

Topology: a star.  Rank 0 listens, every other rank connects once; all collectives are SPMD
(every rank calls them in the same order), so the hub needs no thread: it reads one frame from
every rank, combines, answers.  No population data of the hot path goes through here — halo
planes travel over RCCL (csrc/comm.cpp); only the debugging transport ``HostStagedHalo`` routes
planes through the hub.

Port: rank 0 listens on ``XLB_RDV_PORT`` if set, else on the first free port of
MASTER_PORT, MASTER_PORT + 1, ...  Under ``torch.distributed.run`` MASTER_PORT itself belongs to
the launcher's own store, so the search starts one above it.  Every connection starts with a
job token (hash of the job description), so a foreign listener on a candidate port is skipped.
"""

import atexit
import hashlib
import io
import json
import mmap
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"XLBRDV1\0"
_N_CANDIDATES = 24


def _job_token(addr, port, world, listen_port=None):
    """Hash of the job description; with ``listen_port`` also of the port the hub really bound (a hello meant for the
    hub of one candidate port is refused by a hub that listens on another one)."""
    job = os.environ.get("XLB_JOB_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    tail = "" if listen_port is None else f"|{listen_port}"
    return hashlib.sha256(f"{addr}|{port}|{world}|{job}{tail}".encode()).digest()[:16]


def candidate_ports(master_port):
    """Ports rank 0 may listen on, in the order every rank tries them."""
    fixed = os.environ.get("XLB_RDV_PORT", "").strip()
    if fixed:
        return [int(fixed)]
    under_torchrun = bool(os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("TORCHELASTIC_USE_AGENT_STORE"))
    ports = []
    for k in range(1 if under_torchrun else 0, _N_CANDIDATES):
        p = master_port + k
        ports.append(p if p < 65536 else 20000 + (p - 65536))
    return ports


# ---- frames: 8-byte length + payload ----------------------------------------------------------------
def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)))
    sock.sendall(payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("rendezvous peer closed the connection")
        got += k
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


# ---- values: JSON scalars / containers, raw bytes, NumPy arrays (npy format, no pickle) --------------
def pack(obj):
    if isinstance(obj, np.ndarray):
        bio = io.BytesIO()
        np.save(bio, obj, allow_pickle=False)
        return b"A" + bio.getvalue()
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return b"B" + bytes(obj)
    return b"J" + json.dumps(obj).encode()


def unpack(data):
    tag, body = data[:1], data[1:]
    if tag == b"A":
        return np.load(io.BytesIO(body), allow_pickle=False)
    if tag == b"B":
        return body
    if tag == b"J":
        return json.loads(body.decode())
    raise ValueError("bad rendezvous frame")


def _pack_many(frames):
    """dict {int key: bytes} -> one payload"""
    out = [struct.pack("<I", len(frames))]
    for k, v in frames.items():
        out.append(struct.pack("<iQ", int(k), len(v)))
        out.append(v)
    return b"".join(out)


def _unpack_many(data):
    (n,) = struct.unpack_from("<I", data, 0)
    off, out = 4, {}
    for _ in range(n):
        k, ln = struct.unpack_from("<iQ", data, off)
        off += 12
        out[k] = data[off : off + ln]
        off += ln
    return out


class _Mailbox:
    """A file in /dev/shm mapped into memory: [u64 payload bytes][u32 n]{i32 dst, u64 offset, u64 length} x n, frames.  The owner rewrites
    it per route() call (growing the file when needed); readers map it lazily and re-map when it has grown."""

    _HEAD = 12

    def __init__(self, path, create):
        self.path, self.owner = path, create
        # the owner's file must not exist yet (the name carries a per-job nonce) and must not be a link somebody planted
        flags = os.O_RDWR | os.O_NOFOLLOW | (os.O_CREAT | os.O_EXCL if create else 0)
        self.fd = os.open(path, flags, 0o600)
        self.size = 0
        self.map = None
        if create:
            self._resize(1 << 20)

    def _resize(self, size):
        if self.map is not None:
            self.map.close()
        os.ftruncate(self.fd, size)
        self.map = mmap.mmap(self.fd, size)
        self.size = size

    def _remap(self, need):
        if self.map is None or need > self.size:
            if self.map is not None:
                self.map.close()
            self.size = os.fstat(self.fd).st_size
            self.map = mmap.mmap(self.fd, self.size)

    def write(self, frames):
        """Lay the frames out; returns the number of bytes a reader has to see (0: nothing to read)."""
        if not frames:
            return 0
        table = self._HEAD + 20 * len(frames)
        total = table + sum(len(v) for v in frames.values())
        if total > self.size:
            self._resize(max(total, 2 * self.size))
        m, off = self.map, table
        struct.pack_into("<QI", m, 0, total, len(frames))
        for i, (dst, data) in enumerate(frames.items()):
            struct.pack_into("<iQQ", m, self._HEAD + 20 * i, int(dst), off, len(data))
            m[off : off + len(data)] = data
            off += len(data)
        return total

    def read(self, me, need):
        self._remap(need)
        m = self.map
        total, n = struct.unpack_from("<QI", m, 0)
        for i in range(n):
            dst, off, ln = struct.unpack_from("<iQQ", m, self._HEAD + 20 * i)
            if dst == me:
                return bytes(m[off : off + ln])
        return None

    def close(self, unlink):
        try:
            if self.map is not None:
                self.map.close()
            os.close(self.fd)
            if unlink:
                os.unlink(self.path)
        except OSError:
            pass
        self.map = None


class Rendezvous:
    def __init__(self, rank, world, addr="127.0.0.1", port=29500, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.addr, self.port = addr, int(port)
        self.timeout = float(timeout if timeout is not None else os.environ.get("XLB_RDV_TIMEOUT", 600))
        self._token = _job_token(addr, self.port, self.world)
        self._nonce = ""   # drawn by rank 0 once the star stands: names of this job's /dev/shm files
        self._peers = {}   # hub: rank -> socket
        self._hub = None   # others: socket to rank 0
        self._listener = None
        self._box = None   # this rank's shared-memory mailbox (route through /dev/shm when every rank sits on this host)
        self._box_views = {}
        self._shm = False
        if self.world > 1:
            if self.rank == 0:
                self._serve()
            else:
                self._connect()
            self._shm = self._agree_on_shm()

    # -- set-up
    def _serve(self):
        last = None
        for p in candidate_ports(self.port):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind(("" if self.addr not in ("127.0.0.1", "localhost") else "127.0.0.1", p))
                s.listen(self.world + 8)
                self._listener, self.listen_port = s, p
                break
            except OSError as e:  # in use (e.g. the launcher's own store): next candidate
                last = e
                s.close()
        if self._listener is None:
            raise RuntimeError(f"rendezvous: no free port among {candidate_ports(self.port)}: {last}")
        deadline = time.monotonic() + self.timeout
        self._listener.settimeout(1.0)
        while len(self._peers) < self.world - 1:
            if time.monotonic() > deadline:
                missing = sorted(set(range(1, self.world)) - set(self._peers))
                raise TimeoutError(f"rendezvous: ranks {missing} did not connect to port {self.listen_port} within {self.timeout:.0f} s")
            try:
                conn, _ = self._listener.accept()
            except socket.timeout:
                continue
            try:
                conn.settimeout(10.0)
                hello = _recv_exact(conn, len(_MAGIC) + 16 + 4)
                r = struct.unpack("<i", hello[-4:])[0]
                want = _job_token(self.addr, self.port, self.world, self.listen_port)
                if hello[: len(_MAGIC)] != _MAGIC or hello[len(_MAGIC) : -4] != want or not (0 < r < self.world) or r in self._peers:
                    conn.close()
                    continue
                conn.sendall(b"OK")
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[r] = conn
            except (OSError, ConnectionError):
                conn.close()

    def _connect(self):
        deadline = time.monotonic() + self.timeout
        ports = candidate_ports(self.port)
        while True:
            for p in ports:
                hello = _MAGIC + _job_token(self.addr, self.port, self.world, p) + struct.pack("<i", self.rank)
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(3.0)
                try:
                    s.connect((self.addr, p))
                    s.sendall(hello)
                    s.settimeout(10.0)
                    if _recv_exact(s, 2) == b"OK":
                        s.settimeout(self.timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._hub = s
                        return
                except (OSError, ConnectionError):
                    pass
                s.close()
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {self.rank} found no hub on {self.addr} ports {ports} within {self.timeout:.0f} s")
            time.sleep(0.2)

    # -- primitives
    def all_gather_bytes(self, payload):
        """Every rank's payload, in rank order, on every rank."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = {0: payload}
            for r, s in self._peers.items():
                parts[r] = _recv(s)
            blob = _pack_many(parts)
            for s in self._peers.values():
                _send(s, blob)
        else:
            _send(self._hub, payload)
            parts = _unpack_many(_recv(self._hub))
        return [parts[r] for r in range(self.world)]

    # -- bulk point-to-point through shared memory (one node): the hub only carries the two barriers
    def _box_path(self, r):
        return f"/dev/shm/xlbamd-{self._token.hex()[:12]}-{self._nonce}-{r}"

    def _agree_on_shm(self):
        """True when every rank runs on this host and could create its mailbox file (XLB_RDV_SHM=0 forces the hub path)."""
        ok, ident = False, ""
        # a fresh nonce per job: two jobs started from the same environment must not share (truncate) each other's mailboxes
        self._nonce = self.broadcast(os.urandom(8).hex() if self.rank == 0 else None, src=0)
        try:
            if os.environ.get("XLB_RDV_SHM", "1") != "0" and os.path.isdir("/dev/shm"):
                ident = socket.gethostname() + ":" + open("/proc/sys/kernel/random/boot_id").read().strip()
                self._box = _Mailbox(self._box_path(self.rank), create=True)
                atexit.register(self._drop_box)  # (a job that never calls close() must not leave its file in /dev/shm: it is memory)
                ok = True
        except OSError:
            ok = False
        views = self.all_gather([bool(ok), ident])
        agreed = all(v[0] for v in views) and len({v[1] for v in views}) == 1
        if not agreed and self._box is not None:
            self._box.close(unlink=True)
            self._box = None
        return agreed

    def _drop_box(self):
        if self._box is not None:
            self._box.close(unlink=True)
            self._box = None

    def _route_shm(self, outgoing):
        mine = outgoing.pop(self.rank, None)
        size = self._box.write(outgoing)
        sizes = [struct.unpack("<Q", b)[0] for b in self.all_gather_bytes(struct.pack("<Q", size))]  # barrier 1: every mailbox is written
        got = {} if mine is None else {self.rank: mine}
        for r in range(self.world):
            if r == self.rank or sizes[r] == 0:
                continue
            view = self._box_views.get(r)
            if view is None:
                view = self._box_views[r] = _Mailbox(self._box_path(r), create=False)
            frame = view.read(self.rank, sizes[r])
            if frame is not None:
                got[r] = frame
        self.all_gather_bytes(b"")  # barrier 2: everybody has read; the mailboxes may be rewritten
        return got

    def route(self, outgoing):
        """Point-to-point: ``outgoing`` = {destination rank: bytes}; returns {source rank: bytes}.  Collective: every rank calls it
        (with an empty dict when it has nothing to send).  On one node the payload goes through shared-memory mailboxes and the hub
        only synchronises; otherwise everything goes through the hub."""
        if self.world == 1:
            return {0: outgoing[0]} if 0 in outgoing else {}
        if self._shm:
            return self._route_shm(dict(outgoing))
        if self.rank == 0:
            boxes = {r: {} for r in range(self.world)}
            for dst, data in outgoing.items():
                boxes[dst][0] = data
            for r, s in self._peers.items():
                for dst, data in _unpack_many(_recv(s)).items():
                    boxes[dst][r] = data
            for r, s in self._peers.items():
                _send(s, _pack_many(boxes[r]))
            return boxes[0]
        _send(self._hub, _pack_many(outgoing))
        return _unpack_many(_recv(self._hub))

    # -- conveniences
    def all_gather(self, obj):
        return [unpack(b) for b in self.all_gather_bytes(pack(obj))]

    def broadcast(self, obj, src=0):
        return self.all_gather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather_bytes(b"")

    def all_reduce(self, value, op="max"):
        vals = self.all_gather(float(value))
        return {"max": max, "min": min, "sum": sum}[op](vals)

    def close(self):
        for s in list(self._peers.values()) + [self._hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._hub, self._listener = {}, None, None
        for view in self._box_views.values():
            view.close(unlink=False)
        self._box_views = {}
        if self._box is not None:
            self._box.close(unlink=True)
            self._box = None
        self._shm = False


def from_env(timeout=None):
    """The rendezvous of the job the environment describes (single-process: a trivial one)."""
    rank = int(os.environ.get("RANK", "0") or 0)
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1") or "127.0.0.1"
    port = int(os.environ.get("MASTER_PORT", "29500") or 29500)
    return Rendezvous(rank, world, addr, port, timeout=timeout)
