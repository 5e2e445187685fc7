"""Worker of tests/test_gpu_multirank.py: the full HIP slab path (per-slab masks from global indices,
ghost planes, ring exchange, edge/interior kernels) on WORLD_SIZE ranks, checked against the
single-domain oracle.  XLB_TEST_TRANSPORT=rccl uses the product transport (one GPU per rank);
=host moves the ghost planes through gloo so that all ranks can share ONE GPU (XLB_HIP_DEVICE=0),
which RCCL refuses ("Duplicate GPU detected")."""

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
    rank, world = xdist.init_process_group(transport=transport)
    from _util import hip_cavity_3d

    ok = True
    ctx = xlb_amd.default_config.get_context()
    cases = [(walls_cls, (8 * world + 3, 12, 16), 1) for walls_cls in (HalfwayBounceBackBC, FullwayBounceBackBC)]
    # shapes the two-step kernel takes (fuse2 = 2: no chip-filling rule): pairs of steps with the depth-2 exchange
    cases += [(walls_cls, (18 * world + 1, 8, 64), 2) for walls_cls in (HalfwayBounceBackBC, FullwayBounceBackBC)]
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
    tot = xdist.all_reduce_sum(0.0 if ok else 1.0)
    if rank == 0:
        print("GPU_SLAB_OK" if tot == 0 else "GPU_SLAB_MISMATCH")
    sys.exit(0 if tot == 0 else 1)


if __name__ == "__main__":
    main()
