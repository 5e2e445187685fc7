"""Worker of tests/test_distributed_gloo.py: the slab protocol on CPU over gloo.

Each rank owns an x-slab (+ one ghost plane per side) of a global D3Q19 box, refills its ghosts
every step with the messages of xlb_amd.distribute.SlabPlan (the host description of what
csrc/comm.cpp sends over RCCL) and advances its slab with the oracle.  After K steps the
gathered result must equal the single-domain oracle run bit for bit."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import xlb_numpy as orc  # noqa: E402
from xlb_amd.distribute import SlabPlan  # noqa: E402


def exchange(f_ext, plan):
    """Ring exchange of the face-crossing populations into the ghost planes (gloo p2p)."""
    reqs, recvs = [], []
    for _, pops, send_plane, ghost_plane, send_peer, recv_peer in plan.messages():
        if send_peer is not None:
            buf = torch.from_numpy(np.ascontiguousarray(f_ext[pops, send_plane]))
            reqs.append(dist.isend(buf, dst=send_peer))
        if recv_peer is not None:
            rbuf = torch.empty((len(pops),) + f_ext.shape[2:], dtype=torch.from_numpy(f_ext[:1, 0]).dtype)
            reqs.append(dist.irecv(rbuf, src=recv_peer))
            recvs.append((pops, ghost_plane, rbuf))
    for r in reqs:
        r.wait()
    for pops, ghost_plane, rbuf in recvs:
        f_ext[pops, ghost_plane] = rbuf.numpy()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lattice = os.environ.get("XLB_TEST_LATTICE", "D3Q19")
    walls = os.environ.get("XLB_TEST_WALLS", orc.KIND_HALFWAY_BB)
    steps = 6
    shape = (4 * world + 1, 6, 8)  # uneven split on purpose
    lat, _, bcs = orc.cavity_3d(shape, walls, lattice=lattice)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    f_glob = orc.perturbed_init(shape, lat, seed=13)
    expected = orc.run(f_glob, bc_mask, missing, bcs, 1.3, lat, steps)

    plan = SlabPlan(shape[0], rank, world, lat.c[0])
    x0, nxl = plan.x_offset, plan.nx_local
    ext = lambda a: np.concatenate([np.zeros_like(a[:, :1]), a[:, x0 : x0 + nxl], np.zeros_like(a[:, :1])], axis=1)  # noqa: E731
    f = ext(f_glob)
    bm, mm = ext(bc_mask), ext(missing)
    for _ in range(steps):
        exchange(f, plan)
        new = orc.step(f, bm, mm, bcs, 1.3, lat)
        f[:, 1:-1] = new[:, 1:-1]
    parts = [None] * world
    dist.all_gather_object(parts, f[:, 1:-1])
    got = np.concatenate(parts, axis=1)
    ok = np.array_equal(got, expected)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        print("SLAB_PROTOCOL_OK" if all(flags) else f"SLAB_PROTOCOL_MISMATCH max|d|={np.abs(got - expected).max()}")
    dist.destroy_process_group()
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()
