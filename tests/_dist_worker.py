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


def exchange(f_ext, plan, messages=None):
    """Ring exchange of the face-crossing populations into the ghost planes (gloo p2p)."""
    reqs, recvs = [], []
    for _, pops, send_plane, ghost_plane, send_peer, recv_peer in (plan.messages() if messages is None else messages):
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

    depth = int(os.environ.get("XLB_TEST_DEPTH", "1"))
    plan = SlabPlan(shape[0], rank, world, lat.c[0], halo=depth)
    x0, nxl = plan.x_offset, plan.nx_local
    h = depth
    ghosts = lambda a: np.zeros_like(a[:, :h])  # noqa: E731
    ext = lambda a: np.concatenate([ghosts(a), a[:, x0 : x0 + nxl], ghosts(a)], axis=1)  # noqa: E731
    f = ext(f_glob)
    bm, mm = ext(bc_mask), ext(missing)
    if depth == 1:
        for _ in range(steps):
            exchange(f, plan)
            new = orc.step(f, bm, mm, bcs, 1.3, lat)
            f[:, 1:-1] = new[:, 1:-1]
    else:
        # what xlbhip_run does with two ghost planes: ONE exchange per PAIR of steps (SlabPlan.messages(2)), f(t+1)
        # recomputed on the ghost planes -1 and nx — which takes the neighbours' boundary masks there (mask_messages)
        exchange(bm, plan, plan.mask_messages())
        exchange(mm, plan, [(t, np.arange(lat.q), s_, g, sp, rp) for t, _, s_, g, sp, rp in plan.mask_messages()])
        assert steps % 2 == 0
        for _ in range(steps // 2):
            exchange(f, plan, plan.messages(2))
            mid = orc.step(f, bm, mm, bcs, 1.3, lat)     # valid on planes -1 .. nx (the outermost ghosts wrap onto garbage)
            mid[:, 0] = np.nan                             # poison what must not be used
            mid[:, -1] = np.nan
            new = orc.step(mid, bm, mm, bcs, 1.3, lat)   # valid on planes 0 .. nx - 1
            f[:, h:-h] = new[:, h:-h]
    parts = [None] * world
    dist.all_gather_object(parts, f[:, h:-h])
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
