"""Slab decomposition across GPUs: one process per GPU, ring halo exchange.

Reference semantics: xlb/distribute/distribute.py:18-48 — the domain is split along the
slowest spatial axis (array axis 1) and, every step, the populations with c_x = +1 / -1 of the
slab faces travel to the right / left ring neighbour (``rightPerm`` / ``leftPerm``, periodic
wrap included).  Here each rank keeps ONE ghost x-plane per side, filled before the pull by
RCCL ``ncclSend``/``ncclRecv`` on a dedicated HIP stream (csrc/comm.cpp) and overlapped with the
update of the planes that do not touch a ghost (csrc/api.hip: step_once).

Process-group plumbing only (rendezvous, unique-id broadcast, barriers, gathering results for
tests) goes through ``torch.distributed`` with the ``gloo`` backend; no tensor data of the hot
path touches torch.
"""

import os

import numpy as np

from .. import _lib
from ..default_config import DefaultConfig, get_context
from ..grid.hip_grid import slab_bounds

_state = {"dist": None, "rank": 0, "world": 1}


def _env_int(name, default):
    v = os.environ.get(name, "").strip()
    return int(v) if v else default


def init_process_group(periodic_x=True, init_device_comm=True, transport="rccl"):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
    (as set by ``python -m torch.distributed.run``).  Single-process jobs return (0, 1)
    without importing torch."""
    rank, world = _env_int("RANK", 0), _env_int("WORLD_SIZE", 1)
    _state["rank"], _state["world"] = rank, world
    if world == 1:
        return 0, 1
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    _state["dist"] = dist
    if init_device_comm:
        ctx = get_context()  # device = XLB_HIP_DEVICE or LOCAL_RANK (default_config._pick_device)
        if transport == "rccl":
            box = [_lib.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(rank, world, box[0], periodic_x=periodic_x)
        elif transport == "host":
            # debugging transport: ghost planes travel through host memory and gloo (HostStagedHalo);
            # lets several ranks share ONE GPU, which RCCL refuses ("Duplicate GPU detected")
            ctx.rank, ctx.n_ranks = rank, world
            ctx.set_option("external_halo", 1)
        else:
            raise ValueError(f"unknown halo transport {transport!r}")
    return rank, world


class HostStagedHalo:
    """Ring halo exchange through host memory + gloo, same messages as csrc/comm.cpp (SlabPlan).
    A debugging transport (orders of magnitude slower than RCCL over xGMI); used with the
    ``external_halo`` option: call ``exchange(f)`` before every step."""

    def __init__(self, grid, velocity_set, periodic=True):
        self.plan = SlabPlan(grid.shape[0], grid.rank, grid.n_ranks, velocity_set._c[0], periodic=periodic)

    def exchange(self, f):
        import torch

        dist = _state["dist"]
        reqs, recvs = [], []
        for _, pops, send_plane, ghost_plane, send_peer, recv_peer in self.plan.messages():
            if send_peer is not None:
                buf = torch.from_numpy(np.stack([f.get_plane(int(l), send_plane) for l in pops]))
                reqs.append(dist.isend(buf, dst=send_peer))
            if recv_peer is not None:
                rbuf = torch.from_numpy(np.empty((len(pops),) + tuple(f._s3[1:]), dtype=f.dtype))
                reqs.append(dist.irecv(rbuf, src=recv_peer))
                recvs.append((pops, ghost_plane, rbuf))
        for r in reqs:
            r.wait()
        for pops, ghost_plane, rbuf in recvs:
            arr = rbuf.numpy()
            for i, l in enumerate(pops):
                f.set_plane(int(l), ghost_plane, arr[i])


def rank():
    return _state["rank"]


def world_size():
    return _state["world"]


def barrier():
    if _state["dist"] is not None:
        _state["dist"].barrier()


def all_reduce_max(value):
    if _state["dist"] is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    _state["dist"].all_reduce(t, op=_state["dist"].ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value):
    if _state["dist"] is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    _state["dist"].all_reduce(t, op=_state["dist"].ReduceOp.SUM)
    return float(t.item())


def gather_field(field):
    """All ranks' slabs concatenated along x -> the global (cardinality, nx, ny, nz) array
    (on every rank).  Test / post-processing helper, not on the hot path."""
    local = field.numpy()
    dist = _state["dist"]
    if dist is None:
        return local
    parts = [None] * _state["world"]
    dist.all_gather_object(parts, local)
    return np.concatenate(parts, axis=1)


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

    def __init__(self, nx_global, rank, n_ranks, c_x, periodic=True):
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self.x_offset, self.nx_local = slab_bounds(nx_global, rank, n_ranks)
        c_x = np.asarray(c_x)
        self.right_indices = np.nonzero(c_x == 1)[0]
        self.left_indices = np.nonzero(c_x == -1)[0]
        self.right_rank = (rank + 1) % n_ranks
        self.left_rank = (rank - 1) % n_ranks
        self.has_right = periodic or rank + 1 < n_ranks
        self.has_left = periodic or rank > 0

    def messages(self):
        """[(direction, populations, send storage-plane, recv ghost storage-plane, send peer, recv peer)]
        with storage planes counted INCLUDING the left ghost (interior = 1..nx_local)."""
        nx = self.nx_local
        out = []
        if self.has_right or self.has_left:
            out.append(("right", self.right_indices, nx, 0, self.right_rank if self.has_right else None, self.left_rank if self.has_left else None))
            out.append(("left", self.left_indices, 1, nx + 1, self.left_rank if self.has_left else None, self.right_rank if self.has_right else None))
        return out
