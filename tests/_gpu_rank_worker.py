"""Worker of tests/test_gpu_multirank.py: the full HIP slab path (per-slab masks from global indices,
ghost planes, ring exchange, edge/interior kernels) on WORLD_SIZE ranks, checked against the
single-domain oracle.  XLB_TEST_TRANSPORT=rccl uses RCCL (one GPU per rank); =ipc the IPC-mapped
peer copies on the communication stream (device-to-device, ranks may share ONE GPU: XLB_HIP_DEVICE=0);
=host moves the ghost planes through the host, also on one GPU, which RCCL refuses ("Duplicate GPU detected")."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import xlb_amd  # noqa: E402
from oracle import xlb_numpy as orc  # noqa: E402
from xlb_amd import distribute as xdist  # noqa: E402
from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402


def main():
    transport = os.environ.get("XLB_TEST_TRANSPORT", "rccl")
    # XLB_TEST_CHAIN=1: the ring opened into a chain (rank 0 and rank N-1 exchange nothing) — what bench.py runs the halfway cavity
    # with: its x faces are walls no population is pulled across, so the result is the ring's.  Fullway walls need the ring (their
    # cells exchange — inert — populations with their periodic images) and are left out of a chain run.
    chain = os.environ.get("XLB_TEST_CHAIN", "") == "1"
    rank, world = xdist.init_process_group(transport=transport, periodic_x=not chain)
    from _util import hip_cavity_3d

    ok = True
    ctx = xlb_amd.default_config.get_context()
    wall_kinds = (HalfwayBounceBackBC,) if chain else (HalfwayBounceBackBC, FullwayBounceBackBC)
    cases = [(walls_cls, (8 * world + 3, 12, 16), 1) for walls_cls in wall_kinds]
    # shapes the two-step kernel takes (fuse2 = 2: no chip-filling rule): pairs of steps with the depth-2 exchange
    cases += [(walls_cls, (18 * world + 1, 8, 64), 2) for walls_cls in wall_kinds]
    for walls_cls, shape, fuse2 in cases:
        ctx.set_option("fuse2", fuse2)
        grid, bcs, lat, obcs = hip_cavity_3d(shape, walls_cls)
        assert grid.n_ranks == world and grid.halo == 2
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        x0, nxl = grid.x_offset, grid.local_shape[0]
        ok &= np.array_equal(bc_mask.numpy(), o_bm[:, x0 : x0 + nxl]) and np.array_equal(missing_mask.numpy(), o_mm[:, x0 : x0 + nxl].astype(np.uint8))
        f_np = orc.perturbed_init(shape, lat, seed=5)
        f_0.assign(f_np[:, x0 : x0 + nxl])
        steps = 7
        for overlap in (1, 0):
            xlb_amd.default_config.get_context().set_option("overlap", overlap)
            if fuse2 == 2:
                ok &= stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
            # transport "host": stepper.run drives the same kernels with the ghost planes moved through gloo
            a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.2, steps)
            got = xdist.gather_field(a)
            exp = orc.run(f_np, o_bm, o_mm, obcs, 1.2, lat, steps)
            good = bool(np.array_equal(got, exp))
            if not good and rank == 0:
                print(f"mismatch: {walls_cls.__name__} shape {shape} fuse2 {fuse2} overlap {overlap}", flush=True)
            ok &= good
            f_0.assign(f_np[:, x0 : x0 + nxl])
    # uneven slabs on either side of the two-step kernel's chip-filling rule.  fuse2 = 1 with the rule
    # scaled to a 2-CU chip: 64 planes -> 2 x-segments -> fills it -> eligible; 63 planes -> 1 segment -> half empty -> not.
    # Pairs and single steps post different message sets, so the ranks must agree (MIN over the ranks) or the run hangs.
    if world == 2 and not chain:
        ctx.set_option("fuse2", 1)
        ctx.set_option("fuse2_cus", 2)
        ctx.set_option("overlap", 1)
        shape = (127, 8, 64)
        grid, bcs, lat, obcs = hip_cavity_3d(shape, FullwayBounceBackBC)
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        local = stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        votes = xdist.all_gather(bool(local))
        good = votes == [True, False]
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        x0, nxl = grid.x_offset, grid.local_shape[0]
        f_np = orc.perturbed_init(shape, lat, seed=11)
        f_0.assign(f_np[:, x0 : x0 + nxl])
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.1, 5)
        good &= bool(np.array_equal(xdist.gather_field(a), orc.run(f_np, o_bm, o_mm, obcs, 1.1, lat, 5)))
        if not good and rank == 0:
            print(f"mismatch: slabs straddling the fuse2 rule, votes {votes}", flush=True)
        ok &= good
        ctx.set_option("fuse2_cus", 0)
    if not chain:  # (inlet / outlet faces and fullway walls at the x ends: the ring)
        # widened rows across a slab boundary: profile inlet on rank 0, extrapolation outflow on the last rank, a halfway
        # sphere given by interior indices that straddles the rank boundary, fullway walls (extended kernel variant,
        # k_outflow_aux, per-rank profile table, solid marks in the masker); force on the sphere summed over ranks
        from _util import init_hip
        from xlb_amd.grid import grid_factory
        from xlb_amd.operator.boundary_condition import ExtrapolationOutflowBC, RegularizedBC
        from xlb_amd.operator.force import MomentumTransfer

        ctx.set_option("fuse2", 1)
        ctx.set_option("overlap", 1)
        shape = (14 * world, 12, 12)
        init_hip("D3Q19")
        lat = orc.Lattice("D3Q19")
        grid = grid_factory(shape)
        box, box_ne = orc.bounding_box_indices(shape), orc.bounding_box_indices(shape, remove_edges=True)
        walls = np.unique(np.array([sum((list(box[f][i]) for f in ("bottom", "top", "front", "back")), []) for i in range(3)]), axis=-1)
        x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
        sphere = np.array(np.where((x - shape[0] // 2 + 0.5) ** 2 + (y - 6) ** 2 + (z - 6) ** 2 < 3.1**2))
        yy, zz = np.meshgrid(np.arange(12), np.arange(12), indexing="ij")
        ux = 0.04 * np.maximum(0.0, 1.0 - ((2.0 * (yy - 5.5) / 11.0) ** 2 + (2.0 * (zz - 5.5) / 11.0) ** 2))
        prof = np.stack([ux, np.zeros_like(ux), np.zeros_like(ux)])
        b_w = FullwayBounceBackBC(indices=walls.tolist())
        b_in = RegularizedBC("velocity", profile=lambda: prof, indices=[list(v) for v in box_ne["left"]])
        b_out = ExtrapolationOutflowBC(indices=[list(v) for v in box_ne["right"]])
        b_s = HalfwayBounceBackBC(indices=sphere.tolist())
        obcs = [orc.BC(orc.KIND_FULLWAY_BB, b_w.id, walls), orc.BC(orc.KIND_REGULARIZED_VELOCITY, b_in.id, box_ne["left"], prescribed=prof),
                orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, b_out.id, box_ne["right"]), orc.BC(orc.KIND_HALFWAY_BB, b_s.id, sphere)]
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_in, b_out, b_s])
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        x0, nxl = grid.x_offset, grid.local_shape[0]
        good = np.array_equal(bc_mask.numpy(), o_bm[:, x0 : x0 + nxl]) and np.array_equal(missing_mask.numpy(), o_mm[:, x0 : x0 + nxl].astype(np.uint8))
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.5, 21)
        with np.errstate(all="ignore"):
            exp = orc.run(orc.initialize_eq(shape, lat), o_bm, o_mm, obcs, 1.5, lat, 21)
        good &= bool(np.array_equal(xdist.gather_field(a), exp))
        force = MomentumTransfer(b_s)(a, b, bc_mask, missing_mask)
        ef = orc.momentum_transfer(exp, obcs[3], o_bm, o_mm, lat)
        good &= bool(np.allclose(force, ef, rtol=1e-5, atol=1e-5 * np.abs(ef).max()))
        if not good and rank == 0:
            print("mismatch: sphere channel across ranks", flush=True)
        ok &= good
    if transport in ("ipc", "rccl") or xdist.transport() in ("ipc", "rccl"):
        # device-side transports run the native slab protocol: the overlapped runs above timed the compute stream's halo waits
        stats = ctx.comm_stats()
        good = stats["halo_waits"] > 0
        if not good and rank == 0:
            print(f"telemetry: {stats}", flush=True)
        ok &= good
    tot = xdist.all_reduce_sum(0.0 if ok else 1.0)
    if rank == 0:
        print("GPU_SLAB_OK" if tot == 0 else "GPU_SLAB_MISMATCH")
    sys.exit(0 if tot == 0 else 1)


if __name__ == "__main__":
    main()
