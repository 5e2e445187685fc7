"""Slab decomposition across GPUs: one process per GPU, ring halo exchange.

Reference semantics: xlb/distribute/distribute.py:18-48 — the domain is split along the
slowest spatial axis (array axis 1) and, every step, the populations with c_x = +1 / -1 of the
slab faces travel to the right / left ring neighbour (``rightPerm`` / ``leftPerm``, periodic
wrap included).  Here each rank keeps ghost x-planes, filled before the pull on a dedicated HIP stream
(csrc/comm.cpp) — by RCCL ``ncclSend``/``ncclRecv``, or by copy-engine pulls out of the neighbours' IPC-mapped
fields — and overlapped with the update of the planes that do not touch a ghost (csrc/api.hip: step_once, step_twice).

Process-group plumbing (rendezvous, unique-id broadcast, barriers, small reductions, gathering
results for tests) goes through ``rendezvous.py`` — a star of plain TCP sockets with rank 0 as the
hub, standard library only.  Nothing in this package imports torch.
"""

import os

import numpy as np

from .. import _lib
from ..default_config import get_context
from ..grid.hip_grid import slab_bounds

_state = {"rdv": None, "rank": 0, "world": 1}


def _env_int(name, default):
    v = os.environ.get(name, "").strip()
    return int(v) if v else default


_TRANSPORT_CHAINS = {
    "rccl": ["rccl"],
    "ipc": ["ipc"],
    "host": ["host"],
    "rccl_or_host": ["rccl", "ipc", "host"],  # (round-2 name of "auto")
    "auto": ["rccl", "ipc", "host"],
    "ipc_or_host": ["ipc", "host"],
}


def _try_transport(name, ctx, rdv, rank, world, periodic_x):
    """Bring device transport `name` up on this rank and prove it with a small exchange; returns None or the reason it failed.
    Collective (the same calls in the same order on every rank, whatever fails where)."""
    err = None
    if name == "rccl":
        uid = None
        if rank == 0:
            try:
                uid = _lib.comm_unique_id()
            except Exception as e:  # noqa: BLE001 (whatever RCCL / the loader raises: the other ranks must hear about it)
                err = e
        uid = rdv.broadcast(uid, src=0)
        if uid is not None:
            try:
                ctx.comm_init(rank, world, uid, periodic_x=periodic_x)
            except Exception as e:  # noqa: BLE001
                err = e
        elif err is None:
            err = RuntimeError("rank 0 could not create the RCCL unique id")
    else:  # "ipc"
        # a fresh random name for the control block in /dev/shm, drawn by rank 0
        token = rdv.broadcast(os.urandom(12).hex() if rank == 0 else None, src=0)
        same_host = len({tuple(v) for v in rdv.all_gather(_host_identity())}) == 1
        if not same_host:
            err = RuntimeError("the ipc transport needs every rank on one node")
        else:
            try:
                ctx.comm_init_ipc(rank, world, token, periodic_x=periodic_x)
            except Exception as e:  # noqa: BLE001
                err = e
    # every rank must have a communicator before anyone exchanges (a rank without one would leave the others waiting)
    up = [r for r in rdv.all_gather(None if err is None else f"rank {rank}: {err}") if r]
    if not up:
        probe = None
        try:
            probe = _lib.Field(ctx, 19, _PROBE_SHAPE, _lib.F32, halo=2)
            _verify_exchange(ctx, probe, rank, world, periodic_x)
        except Exception as e:  # noqa: BLE001
            err = e
        up = [r for r in rdv.all_gather(None if err is None else f"rank {rank}: {name} self-check: {err}") if r]
        if probe is not None:  # (only now: the neighbours have finished reading it)
            probe.free()
    if up:
        try:
            ctx.comm_destroy()
        except Exception:  # noqa: BLE001
            pass
        ctx.rank, ctx.n_ranks = 0, 1
        return up[0]
    return None


def _host_identity():
    import socket

    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return [socket.gethostname(), boot]


_PROBE_SHAPE = (4, 8, 64)


def _verify_exchange(ctx, f, rank, world, periodic_x):
    """One depth-2 exchange of a small rank-stamped field through the transport that was just set up, ghost planes checked
    against what the neighbours must have sent: a transport that comes up but moves wrong bytes is not trusted with a run."""
    (nx, ny, nz), q = _PROBE_SHAPE, 19

    def stamp(r, l, x):  # value of every cell of interior plane x, population l, on rank r
        return float(1 + r * 1000 + l * 10 + x)

    host = np.empty((q, nx, ny, nz), np.float32)
    for l in range(q):
        for x in range(nx):
            host[l, x] = stamp(rank, l, x)
    f.assign(host)
    ctx.sync()
    _lib.check(_lib.load().xlbhip_halo_exchange_wide(ctx.handle, _lib.D3Q19, f.handle))
    ctx.sync()
    c_x = _lib.lattice_info(_lib.D3Q19)[2][0]
    left, right = (rank - 1) % world, (rank + 1) % world
    has_left, has_right = periodic_x or rank > 0, periodic_x or rank + 1 < world
    for l in range(q):
        checks = []
        if has_left:
            checks.append((1, stamp(left, l, nx - 1)))        # ghost -1  <- left neighbour's last plane, every population
            if c_x[l] == 1:
                checks.append((0, stamp(left, l, nx - 2)))    # ghost -2  <- the plane behind it, populations moving right
        if has_right:
            checks.append((nx + 2, stamp(right, l, 0)))
            if c_x[l] == -1:
                checks.append((nx + 3, stamp(right, l, 1)))
        for storage_plane, want in checks:
            got = f.get_plane(l, storage_plane)
            if not np.all(got == np.float32(want)):
                raise RuntimeError(f"ghost plane {storage_plane} of population {l} holds {got.flat[0]!r}, expected {want!r}")


def init_process_group(periodic_x=True, init_device_comm=True, transport="rccl"):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (as set by
    ``python -m torch.distributed.run`` or by ``bench.py``'s launcher).  Single-process jobs return (0, 1).

    ``periodic_x=False``: the global x axis ends in walls, so the ring is a chain — rank 0 and rank N-1
    exchange nothing (a walled cavity never pulls across that face).

    ``transport``: how ghost planes travel.
      * "rccl": RCCL send / recv (csrc/comm.cpp); an error if the communicator cannot be built or fails its self-check;
      * "ipc": the neighbours' fields mapped through HIP IPC and pulled with copy-engine copies on the communication
        stream, ordered by counters in a host shared-memory block (one node; the ranks may share a device);
      * "host": through host memory and the rendezvous (debugging; orders of magnitude slower);
      * "auto" (= "rccl_or_host", the round-2 name): RCCL, else ipc, else host — a transport that FAILS to come up or
        fails its self-check on any rank is dropped by all ranks together, said on stderr, reported by ``transport()``.
        A hang inside RCCL cannot be caught."""
    rank, world = _env_int("RANK", 0), _env_int("WORLD_SIZE", 1)
    _state["rank"], _state["world"] = rank, world
    if world == 1:
        return 0, 1
    if _state["rdv"] is None:
        from . import rendezvous

        _state["rdv"] = rendezvous.from_env()
    rdv = _state["rdv"]
    _state["periodic_x"] = bool(periodic_x)
    if init_device_comm:
        ctx = get_context()  # device = XLB_HIP_DEVICE or LOCAL_RANK (default_config._pick_device)
        if transport not in _TRANSPORT_CHAINS:
            raise ValueError(f"unknown halo transport {transport!r}")
        chain = _TRANSPORT_CHAINS[transport]
        chosen, why = None, []
        for name in chain:
            if name == "host":
                chosen = "host"
                break
            reason = _try_transport(name, ctx, rdv, rank, world, periodic_x)
            if reason is None:
                chosen = name
                break
            why.append(f"{name}: {reason}")
            if name == chain[-1]:
                raise RuntimeError(f"halo transport {name!r} failed: {reason}")
            if rank == 0:
                import sys

                print(f"xlb_amd.distribute: halo transport {name!r} failed ({reason}); every rank falls back to {chain[chain.index(name) + 1]!r}",
                      file=sys.stderr, flush=True)
        _state["transport"] = chosen if not why else f"{chosen} (fallback: {'; '.join(w[:200] for w in why)})"
        if chosen == "host":
            # ghost planes travel through host memory and the rendezvous hub (HostStagedHalo): the debugging transport and the
            # last resort of "auto"
            ctx.rank, ctx.n_ranks = rank, world
            ctx.set_option("external_halo", 1)
    return rank, world


def transport():
    """The halo transport in use: "rccl", "ipc", "host", or e.g. "ipc (fallback: rccl: <why it failed>)"; None before init_process_group."""
    return _state.get("transport")


def shutdown():
    """Leave the job: tear the device communicator down (the ipc transport tells its neighbours and gives them a moment to finish
    pulling before this rank's memory goes away), then close the control-plane sockets."""
    t = _state.get("transport") or ""
    if t.startswith(("rccl", "ipc")):
        try:
            ctx = get_context()
            ctx.sync()
            ctx.comm_destroy()
            ctx.rank, ctx.n_ranks = 0, 1
        except Exception:  # noqa: BLE001 (leaving anyway)
            pass
    if _state["rdv"] is not None:
        _state["rdv"].close()
    _state.update(rdv=None, rank=0, world=1, transport=None)


class HostStagedHalo:
    """Ring halo exchange through host memory + the rendezvous hub, same messages as csrc/comm.cpp (SlabPlan).
    A debugging transport (orders of magnitude slower than RCCL over xGMI); used with the
    ``external_halo`` option: call ``exchange(f)`` before every step."""

    def __init__(self, grid, velocity_set, periodic=None):
        periodic = _state.get("periodic_x", True) if periodic is None else periodic
        self.plan = SlabPlan(grid.shape[0], grid.rank, grid.n_ranks, velocity_set._c[0], periodic=periodic, halo=grid.halo)

    def exchange(self, f, depth=1):
        """Refill f's ghost planes: depth 1 before a single step, depth 2 before a fused pair of steps."""
        self._run(f, self.plan.messages(depth))

    def exchange_masks(self, bc_mask, missing_mask=None):
        """Ghost planes -1 and nx of the masks (the two-step kernel evaluates boundary conditions there)."""
        self._run(bc_mask, self.plan.mask_messages())
        if missing_mask is not None:
            self._run(missing_mask, self.plan.mask_messages())

    def _run(self, f, messages):
        from .rendezvous import pack, unpack

        rdv = _state["rdv"]
        # one frame per peer: the planes of all messages to that peer, in message order (both sides walk the same list)
        out, expect = {}, []
        for i, (_, pops, send_plane, ghost_plane, send_peer, recv_peer) in enumerate(messages):
            if send_peer is not None:
                out.setdefault(send_peer, []).append((i, np.stack([f.get_plane(int(l), send_plane) for l in pops])))
            if recv_peer is not None:
                expect.append((recv_peer, pops, ghost_plane))
        frames = {}
        for peer, items in out.items():
            frames[peer] = pack(np.concatenate([a.reshape(-1) for _, a in items]))
        got = {src: unpack(b) for src, b in rdv.route(frames).items()}
        cursor = {src: 0 for src in got}
        plane = int(np.prod(f._s3[1:]))
        for src, pops, ghost_plane in expect:
            n = len(pops) * plane
            arr = got[src][cursor[src] : cursor[src] + n].reshape((len(pops),) + tuple(f._s3[1:]))
            cursor[src] += n
            for i, l in enumerate(pops):
                f.set_plane(int(l), ghost_plane, arr[i])


def rank():
    return _state["rank"]


def world_size():
    return _state["world"]


def barrier():
    if _state["rdv"] is not None:
        _state["rdv"].barrier()


def all_reduce_max(value):
    return float(value) if _state["rdv"] is None else _state["rdv"].all_reduce(value, "max")


def all_reduce_min(value):
    return float(value) if _state["rdv"] is None else _state["rdv"].all_reduce(value, "min")


def all_reduce_sum(value):
    return float(value) if _state["rdv"] is None else _state["rdv"].all_reduce(value, "sum")


def all_gather(obj):
    """Every rank's object (JSON-able value, bytes or NumPy array) in rank order."""
    return [obj] if _state["rdv"] is None else _state["rdv"].all_gather(obj)


def gather_field(field):
    """All ranks' slabs concatenated along x -> the global (cardinality, nx, ny, nz) array
    (on every rank).  Test / post-processing helper, not on the hot path."""
    local = field.numpy()
    if _state["rdv"] is None:
        return local
    return np.concatenate(_state["rdv"].all_gather(local), axis=1)


def distribute(operator, grid, velocity_set, num_results=1, ops="permute"):
    """API twin of the reference's ``distribute`` (distribute.py:82-105).  On this backend the
    decomposition lives in the grid (``grid_factory`` hands every rank its slab once
    :func:`init_process_group` ran) and the exchange is part of the native step, so the
    operator is returned unchanged."""
    if ops != "permute":
        raise NotImplementedError(f"Operation {ops} not implemented")
    return operator


class SlabPlan:
    """Host description of one rank's part of the ring exchange: which populations of which
    x-plane go to which neighbour, and into which ghost plane they land.  Mirrors what
    csrc/comm.cpp does; used by the CPU (gloo) protocol tests."""

    def __init__(self, nx_global, rank, n_ranks, c_x, periodic=True, halo=1):
        self.rank, self.n_ranks, self.halo = int(rank), int(n_ranks), int(halo)
        self.q = len(c_x)
        self.x_offset, self.nx_local = slab_bounds(nx_global, rank, n_ranks)
        c_x = np.asarray(c_x)
        self.right_indices = np.nonzero(c_x == 1)[0]
        self.left_indices = np.nonzero(c_x == -1)[0]
        self.right_rank = (rank + 1) % n_ranks
        self.left_rank = (rank - 1) % n_ranks
        self.has_right = periodic or rank + 1 < n_ranks
        self.has_left = periodic or rank > 0

    def _pair(self, tag, pops, send_plane, ghost_plane, to_right):
        """storage planes count the left ghosts: interior plane X lives at X + halo"""
        h = self.halo
        if to_right:
            return (tag, pops, send_plane + h, ghost_plane + h, self.right_rank if self.has_right else None, self.left_rank if self.has_left else None)
        return (tag, pops, send_plane + h, ghost_plane + h, self.left_rank if self.has_left else None, self.right_rank if self.has_right else None)

    def messages(self, depth=1):
        """[(direction, populations, send storage-plane, recv ghost storage-plane, send peer, recv peer)].

        depth 1 (one step): the face-crossing populations of the edge planes.  depth 2 (two fused steps, fields with
        two ghost planes): f(t+1) is recomputed on the ghost planes -1 and nx, which takes EVERY population of the
        neighbour's edge plane and the crossing ones of the plane behind it."""
        nx = self.nx_local
        if depth not in (1, 2) or depth > self.halo:
            raise ValueError(f"halo depth {depth} with {self.halo} ghost plane(s)")
        if not (self.has_right or self.has_left):
            return []
        if depth == 1:
            return [self._pair("right", self.right_indices, nx - 1, -1, True), self._pair("left", self.left_indices, 0, nx, False)]
        every = np.arange(self.q)
        return [
            self._pair("right", every, nx - 1, -1, True),
            self._pair("right2", self.right_indices, nx - 2, -2, True),
            self._pair("left", every, 0, nx, False),
            self._pair("left2", self.left_indices, 1, nx + 1, False),
        ]

    def mask_messages(self):
        """Ghost planes -1 and nx of a per-cell mask (one device plane)."""
        nx = self.nx_local
        if not (self.has_right or self.has_left):
            return []
        return [self._pair("right", np.arange(1), nx - 1, -1, True), self._pair("left", np.arange(1), 0, nx, False)]
