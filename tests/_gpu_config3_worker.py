"""Worker of tests/test_gpu_multirank.py::test_config3_global_domain_decomposed_matches_one_rank.

BASELINE configs[3]'s GLOBAL domain — 4096 x 512 x 512, D3Q19 BGK fp32, lid-driven cavity with halfway walls — advanced 7 steps
(three fused pairs + a single step) from a non-trivial state, (a) by ONE rank, which writes a set of x-planes to a file, and (b)
slab-decomposed over WORLD_SIZE ranks sharing the device over the ipc transport, where every rank compares the planes it owns
with that file: walls, the planes on either side of every rank boundary, planes in between.  Bit for bit."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import xlb_amd  # noqa: E402
from bench import cavity_bcs  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd import distribute as xdist  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.boundary_condition import EquilibriumBC, HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.equilibrium import QuadraticEquilibrium  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402

NX, N, STEPS, OMEGA = int(os.environ.get("XLB_C3_NX", "4096")), int(os.environ.get("XLB_C3_N", "512")), 7, 1.0
POPS = (0, 2, 9, 13, 14, 18)


def planes_to_check(world):
    xs = {0, 1, 2, NX - 3, NX - 2, NX - 1, NX // 2 + 77}
    for w in (2, 3, 4, 5, 8):  # rank boundaries of the decompositions this test may be run with
        base, rem = divmod(NX, w)
        b = 0
        for r in range(w - 1):
            b += base + (1 if r < rem else 0)
            xs.update({b - 2, b - 1, b, b + 1})
    return sorted(x for x in xs if 0 <= x < NX)


def main():
    path = os.environ["XLB_C3_FILE"]
    rank, world = xdist.init_process_group(periodic_x=False, transport=os.environ.get("XLB_TEST_TRANSPORT", "ipc"))
    pp = PrecisionPolicy.FP32FP32
    vs = xlb_amd.velocity_set.D3Q19(precision_policy=pp, compute_backend=ComputeBackend.HIP)
    xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
    ctx = get_context()
    grid = grid_factory((NX, N, N))
    x0, nxl = grid.x_offset, grid.local_shape[0]
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=cavity_bcs(grid, HalfwayBounceBackBC, EquilibriumBC))
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    # a non-trivial state, the same global field whatever the decomposition: f = feq(1 + 0.01 xi, 0.01 eta) on a 32^3 pattern tiled from
    # GLOBAL index 0, evaluated by the backend's own equilibrium operator
    rng = np.random.default_rng(3)
    rho_t = (1.0 + 0.01 * rng.uniform(-1, 1, (1, 32, 32, 32))).astype(np.float32)
    u_t = (0.01 * rng.uniform(-1, 1, (3, 32, 32, 32))).astype(np.float32)
    ix = (x0 + np.arange(nxl)) % 32
    reps = N // 32
    rho = grid.create_field(1, dtype=pp.compute_precision).assign(np.tile(rho_t[:, ix], (1, 1, reps, reps)))
    u = grid.create_field(3, dtype=pp.compute_precision).assign(np.tile(u_t[:, ix], (1, 1, reps, reps)))
    QuadraticEquilibrium()(rho, u, f_0)
    ctx.sync()
    rho.free()
    u.free()
    if world == 1:
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    xdist.barrier()
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, OMEGA, STEPS)
    ctx.sync()
    h = grid.halo
    mine = [x for x in planes_to_check(world) if x0 <= x < x0 + nxl]
    got = {f"{x}_{l}": a.get_plane(l, x - x0 + h) for x in mine for l in POPS}
    if world == 1:
        np.savez(path, **got)
        ref = got["0_0"]
        assert np.isfinite(ref).all() and float(np.abs(got[f"{NX // 2 + 77}_9"] - got[f"{NX // 2 + 77}_9"][0, 0]).max()) > 1e-4  # not a uniform state
        print(f"CONFIG3_REF_OK {len(got)} planes", flush=True)
        return
    ref = np.load(path)
    bad = [k for k, v in got.items() if not np.array_equal(v, ref[k])]
    n_bad = xdist.all_reduce_sum(float(len(bad)))
    n_all = xdist.all_reduce_sum(float(len(got)))
    if bad:
        print(f"rank {rank}: planes differ: {bad[:8]}", flush=True)
    if rank == 0:
        print(("CONFIG3_DECOMPOSED_OK" if n_bad == 0 else "CONFIG3_DECOMPOSED_MISMATCH") + f" {int(n_all)} planes compared on {world} ranks, transport {xdist.transport()}", flush=True)
    xdist.barrier()
    sys.exit(0 if n_bad == 0 else 1)


if __name__ == "__main__":
    main()
