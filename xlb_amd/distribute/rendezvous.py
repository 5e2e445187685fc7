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

import hashlib
import io
import json
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"XLBRDV1\0"
_N_CANDIDATES = 24


def _job_token(addr, port, world):
    job = os.environ.get("XLB_JOB_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    return hashlib.sha256(f"{addr}|{port}|{world}|{job}".encode()).digest()[:16]


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


class Rendezvous:
    def __init__(self, rank, world, addr="127.0.0.1", port=29500, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.addr, self.port = addr, int(port)
        self.timeout = float(timeout if timeout is not None else os.environ.get("XLB_RDV_TIMEOUT", 600))
        self._token = _job_token(addr, self.port, self.world)
        self._peers = {}   # hub: rank -> socket
        self._hub = None   # others: socket to rank 0
        self._listener = None
        if self.world > 1:
            if self.rank == 0:
                self._serve()
            else:
                self._connect()

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
                if hello[: len(_MAGIC)] != _MAGIC or hello[len(_MAGIC) : -4] != self._token or not (0 < r < self.world) or r in self._peers:
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
        hello = _MAGIC + self._token + struct.pack("<i", self.rank)
        ports = candidate_ports(self.port)
        while True:
            for p in ports:
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

    def route(self, outgoing):
        """Point-to-point through the hub: ``outgoing`` = {destination rank: bytes}; returns {source rank: bytes}.
        Collective: every rank calls it (with an empty dict when it has nothing to send)."""
        if self.world == 1:
            return {0: outgoing[0]} if 0 in outgoing else {}
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


def from_env(timeout=None):
    """The rendezvous of the job the environment describes (single-process: a trivial one)."""
    rank = int(os.environ.get("RANK", "0") or 0)
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1") or "127.0.0.1"
    port = int(os.environ.get("MASTER_PORT", "29500") or 29500)
    return Rendezvous(rank, world, addr, port, timeout=timeout)
