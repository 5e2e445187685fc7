"""Full-size checks (BASELINE.json sizes) through size-independent properties, since the
oracle cannot run 512^3 in seconds:
  * tiling: a periodic domain initialised with a 16^3-periodic pattern must stay the tiling of
    the 16^3 oracle solution, bit for bit (exercises every address computation at full size);
  * mass conservation in the periodic box;
  * cavity at full size: mirror symmetry in y of rho / u_x / u_z, antisymmetry of u_y (to rounding), and
    agreement with the committed 16^3 golden masks' structure (counts scale as 6 faces).
"""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, HalfwayBounceBackBC
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import hip_cavity_3d, hip_macroscopic, init_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [256, 512])
def test_periodic_tiling_and_mass(n):
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    t = 16
    tile = orc.perturbed_init((t, t, t), lat, seed=21)
    steps = 6
    bm = np.zeros((1, t, t, t), np.uint8)
    mm = np.zeros((lat.q, t, t, t), bool)
    exp_tile = orc.run(tile, bm, mm, [], 1.7, lat, steps)
    o_rho, o_u = orc.macroscopic(exp_tile, lat)
    grid = grid_factory((n, n, n))
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    reps = n // t
    f_0.assign(np.tile(tile, (1, reps, reps, reps)))
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.7, steps)
    rho, u = hip_macroscopic(f_0, grid, vs, pp)
    assert np.array_equal(rho, np.tile(o_rho, (1, reps, reps, reps)))
    assert np.array_equal(u, np.tile(o_u, (1, reps, reps, reps)))
    mass0 = float(tile.astype(np.float64).sum()) * reps**3
    assert abs(float(rho.astype(np.float64).sum()) - mass0) / mass0 < 1e-6


@pytest.mark.parametrize("walls_cls", [HalfwayBounceBackBC, FullwayBounceBackBC])
def test_cavity_512_symmetry(walls_cls):
    n = 512
    grid, bcs, lat, obcs = hip_cavity_3d((n, n, n), walls_cls)
    vs, pp = bcs[0].velocity_set, bcs[0].precision_policy
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    bm = bc_mask.numpy()
    assert int((bm == bcs[0].id).sum()) == (n - 2) ** 2  # lid = top face without edges
    assert int((bm != 0).sum()) == n**3 - (n - 2) ** 3 - 0  # whole hull is tagged (lid + walls)
    del bm
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, 20)
    rho, u = hip_macroscopic(f_0, grid, vs, pp)
    assert np.isfinite(rho).all() and np.isfinite(u).all()
    # lid moves along +x: the flow is mirror-symmetric about the y mid-plane — to rounding only, because the
    # moment sums run in a fixed direction order that the mirror image permutes
    sym = 2e-6
    assert np.abs(rho - rho[:, :, ::-1, :]).max() <= sym
    assert np.abs(u[0] - u[0][:, ::-1, :]).max() <= sym and np.abs(u[2] - u[2][:, ::-1, :]).max() <= sym
    assert np.abs(u[1] + u[1][:, ::-1, :]).max() <= sym
    assert float(np.abs(u[0]).max()) > 1e-3  # the lid actually drives the flow


@pytest.mark.parametrize("walls_cls", [HalfwayBounceBackBC, FullwayBounceBackBC])
def test_cavity_512_two_step_kernel_equals_single_step_kernel(walls_cls):
    """BASELINE configs[2] at full size: 120 steps through the two-steps-per-pass kernel (hand-counted vmcnt, inline-asm
    fix-up loads, 4096 work items) and through the single-step kernel give the SAME BITS in all 19 x 512^3 populations.
    (The small-size tests compare both kernels with the oracle; this one exercises the full-size schedule.)"""
    from xlb_amd.default_config import get_context

    n, steps = 512, 120
    grid, bcs, lat, obcs = hip_cavity_3d((n, n, n), walls_cls)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    ctx = get_context()
    assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    digests = []
    try:
        for fuse2 in (1, 0):
            ctx.set_option("fuse2", fuse2)
            # the driver's start (f = w everywhere), identical for both runs
            init = orc.initialize_eq((1, 1, 1), lat).reshape(19)
            f_0.assign(np.broadcast_to(init.reshape(19, 1, 1, 1), (19, n, n, n)).astype(np.float32))
            a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, steps)
            out = a.numpy()
            assert np.isfinite(out).all()
            digests.append(out)
            f_0, f_1 = (a, b) if a is f_0 else (b, a)
    finally:
        ctx.set_option("fuse2", 1)
    assert np.array_equal(digests[0], digests[1])
    assert float(np.abs(digests[0] - digests[0][:, :1, :1, :1]).max()) > 1e-4  # the lid drove a flow: not a trivial comparison


def test_cavity_512_slab_layout_equals_plain_layout():
    """The multi-rank field layout (two ghost planes, depth-2 ring exchange onto the rank itself, interior launch
    overlapped with the exchange + two edge launches) against the plain layout at BASELINE configs[2] size: same bits
    after 41 steps (20 fused pairs + one single step with the depth-1 exchange)."""
    n, steps = 512, 41
    outs = []
    for cfg in (None, {"halo": 2}):
        grid, bcs, lat, obcs = hip_cavity_3d((n, n, n), HalfwayBounceBackBC, backend_config=cfg)
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, steps)
        outs.append(a.numpy())
        for fld in (f_0, f_1, bc_mask, missing_mask):
            fld.free()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("exact", [0, 1])
def test_config5_fullsize_d3q27_kbc_fp64fp32_tiling(exact):
    """BASELINE configs[4] at full size (D3Q27 KBC, fp64 compute / fp32 store, 384^3): a 12^3-periodic pattern tiled over
    the domain must stay the tiling of the 12^3 oracle solution — exercises the fp64 kernel's addressing at full size.
    Bit for bit with exact_math=1; within the north-star tolerance (in fact to rounding) with the default fast collision."""
    from xlb_amd.default_config import get_context

    n, t, steps, omega = 384, 12, 6, 1.9
    vs, pp = init_hip("D3Q27", "FP64FP32")
    ctx = get_context()
    lat = orc.Lattice("D3Q27")
    tile = orc.perturbed_init((t, t, t), lat, "FP64FP32", seed=17, amp_rho=0.02, amp_u=0.03)
    bm, mm = np.zeros((1, t, t, t), np.uint8), np.zeros((lat.q, t, t, t), bool)
    exp_tile = orc.run(tile, bm, mm, [], omega, lat, steps, "FP64FP32", "KBC")
    reps = n // t
    try:
        ctx.set_option("exact_math", exact)
        grid = grid_factory((n, n, n))
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[], collision_type="KBC")
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        f_0.assign(np.tile(tile, (1, reps, reps, reps)))
        f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
        out = f_0.numpy()
    finally:
        ctx.set_option("exact_math", 0)
    assert out.dtype == np.float32
    # every 12^3 block of the result equals block (0, 0, 0) bit for bit (the kernel treats all cells alike) ...
    blocks = out.reshape(lat.q, reps, t, reps, t, reps, t)
    assert np.array_equal(blocks, np.broadcast_to(blocks[:, :1, :, :1, :, :1, :], blocks.shape))
    # ... and that block is the oracle's solution
    first = np.ascontiguousarray(blocks[:, 0, :, 0, :, 0, :])
    if exact:
        assert np.array_equal(first, exp_tile)
    else:
        assert np.abs(first.astype(np.float64) - exp_tile.astype(np.float64)).max() <= 1e-6


def test_d3q27_two_step_kernel_fullsize_tiling():
    """D3Q27 BGK fp32 at 384^3 through the two-step kernel (lifetime-packed LDS ring, 2304 work items): the tiled 12^3
    oracle solution, bit for bit, for an odd step count (pairs + one single step)."""
    n, t, steps, omega = 384, 12, 7, 1.6
    vs, pp = init_hip("D3Q27")
    lat = orc.Lattice("D3Q27")
    tile = orc.perturbed_init((t, t, t), lat, seed=19)
    bm, mm = np.zeros((1, t, t, t), np.uint8), np.zeros((lat.q, t, t, t), bool)
    exp_tile = orc.run(tile, bm, mm, [], omega, lat, steps)
    reps = n // t
    grid = grid_factory((n, n, n))
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
    f_0.assign(np.tile(tile, (1, reps, reps, reps)))
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    assert np.array_equal(f_0.numpy(), np.tile(exp_tile, (1, reps, reps, reps)))


def test_config3_domain_on_one_gpu_matches_the_cube_near_its_walls():
    """BASELINE configs[3]'s global domain, 4096 x 512 x 512 (1.07 G cells, 81.6 GB per population field: element offsets beyond 2^34,
    the largest size the path is asked for) on ONE GPU through the two-step kernel, lid-driven cavity with halfway walls.  Information
    travels one cell per step, so after 6 steps the planes near either x wall are those of the 512^3 cavity, bit for bit — checked on
    planes of several populations at both ends — and the middle of the long box, which no x wall has reached, repeats along x."""
    import gc

    from bench import cavity_bcs
    from xlb_amd.default_config import get_context
    from xlb_amd.operator.boundary_condition import EquilibriumBC

    n, nx, steps = 512, 4096, 6
    pops = (0, 2, 9, 13, 14, 18)
    near = (0, 1, 2, 3, 7, 100, 300)
    init_hip("D3Q19")
    ctx = get_context()
    got = {}
    for length in (nx, n):
        grid = grid_factory((length, n, n))
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=cavity_bcs(grid, HalfwayBounceBackBC, EquilibriumBC))
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, steps)
        ctx.sync()
        got[length] = {("lo", x, l): a.get_plane(l, x) for x in near for l in pops}
        got[length].update({("hi", x, l): a.get_plane(l, length - 1 - x) for x in near for l in pops})
        if length == nx:
            mid = {(x, l): a.get_plane(l, x) for x in (2000, 2001, 2047, 2048, 3000) for l in pops}
        for fld in (f_0, f_1, bc_mask, missing_mask):
            fld.free()
        del stepper, f_0, f_1, bc_mask, missing_mask, a, b
        gc.collect()
    for key in got[n]:
        assert np.array_equal(got[nx][key], got[n][key]), key
    for l in pops:
        for x in (2001, 2047, 2048, 3000):
            assert np.array_equal(mid[(x, l)], mid[(2000, l)]), (x, l)
    assert any(np.abs(got[nx][("lo", 3, l)] - got[nx][("lo", 3, l)][0, 0]).max() > 0 for l in pops)  # not a trivial (uniform) comparison


def hull_masks(shape, lat, lid_id, walls_id):
    """The masks of the reference drivers' cavity (mlups_3d.py:193-204) written down directly from what the JAX masker produces
    for hull-only BCs (missing[l, x] <=> x - c_l lies outside the box, for EVERY cell; bc_mask = lid on the
    top face without its edges, walls on the rest of the hull) — O(N) NumPy, affordable at 512^3 where orc.build_masks' padded rolls
    are not.  Checked against orc.build_masks at a small size by the caller."""
    nx, ny, nz = shape
    bm = np.zeros((1, nx, ny, nz), np.uint8)
    b = bm[0]
    b[0], b[-1], b[:, 0], b[:, -1], b[:, :, 0], b[:, :, -1] = (walls_id,) * 6
    b[1:-1, 1:-1, -1] = lid_id
    mm = np.zeros((lat.q, nx, ny, nz), bool)
    for l in range(lat.q):
        for axis, n in enumerate(shape):
            c = int(lat.c[axis, l])
            if c:
                idx = [slice(None)] * 3
                idx[axis] = 0 if c > 0 else n - 1  # x - c < 0 at x = 0 for c = +1; x - c > n - 1 at x = n - 1 for c = -1
                mm[(l,) + tuple(idx)] = True
    return bm, mm


CASES_FULL_ORACLE = [(512, HalfwayBounceBackBC, 4, "tile"), (512, FullwayBounceBackBC, 4, "tile"), (256, HalfwayBounceBackBC, 12, "rest")]


@pytest.mark.parametrize("n, walls_cls, steps, start", CASES_FULL_ORACLE)
def test_cavity_fullsize_against_the_c_oracle(n, walls_cls, steps, start):
    """BASELINE configs[2] at FULL size, compared DIRECTLY with the checker: oracle/lbm_ref.c — the plain-C
    restatement of nse_stepper.py:237-282, bit-identical to the NumPy oracle (tests/test_oracle_c.py) — runs the 512^3 lid-driven
    cavity with halfway and with fullway walls on the host's cores, the HIP stepper runs it through `stepper.run` with default
    options (the two-step kernel), and ALL 19 x n^3 populations must be equal bit for bit.  "tile": the start is a perturbed
    equilibrium (a 64^3 pattern tiled over the box), so every cell — every wall cell, every corner — carries a non-trivial state
    from step one; "rest": the drivers' own start f = w, 12 steps at 256^3."""
    from oracle import lbm_ref

    shape = (n, n, n)
    grid, bcs, lat, obcs = hip_cavity_3d(shape, walls_cls)
    # masks: analytic hull masks, pinned to the oracle's masker at 16^3
    s_bm, s_mm = hull_masks((16, 16, 16), lat, bcs[0].id, bcs[1].id)
    o_small = orc.cavity_3d(16, obcs[1].kind)
    remap = {o_small[2][0].id: bcs[0].id, o_small[2][1].id: bcs[1].id}
    r_bm, r_mm = orc.build_masks((16, 16, 16), lat, o_small[2])
    from _util import remap_ids

    assert np.array_equal(s_bm, remap_ids(r_bm, remap)) and np.array_equal(s_mm, r_mm)
    o_bm, o_mm = hull_masks(shape, lat, bcs[0].id, bcs[1].id)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    assert np.array_equal(bc_mask.numpy(), o_bm)
    assert np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    if start == "tile":
        t = 64
        f_np = np.tile(orc.perturbed_init((t, t, t), lat, seed=77), (1, n // t, n // t, n // t))
        f_0.assign(f_np)
    else:
        f_np = np.ascontiguousarray(np.broadcast_to(orc.initialize_eq((1, 1, 1), lat).reshape(19, 1, 1, 1), (19, n, n, n)).astype(np.float32))
    if n == 512:
        assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)  # the kernel the bench line is quoted on
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, steps)
    got = a.numpy()
    for fld in (f_0, f_1, bc_mask, missing_mask):
        fld.free()
    exp = lbm_ref.run(f_np, o_bm, o_mm, obcs, 1.0, lat, steps)
    del f_np
    assert got.shape == exp.shape == (19, n, n, n) and got.dtype == exp.dtype == np.float32
    assert np.array_equal(got, exp)
    assert float(np.abs(exp[:, 1:-1, 1:-1, -2] - exp[:, 1:2, 1:2, 1:2].reshape(19, 1, 1)).max()) > 1e-4  # the lid drove the layer below it
