"""The tolerance-graded fast collision (cell.hpp: kbc_fast, the default for fp64-compute D3Q27 KBC — BASELINE configs[4]):
within the north-star tolerance (1e-6) of the oracle / golden vectors, while `exact_math=1` selects the bit-exact build.
Reference: xlb/operator/collision/kbc.py:58-94."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.default_config import get_context
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import golden, hip_cavity_3d, init_hip

pytestmark = pytest.mark.gpu
TOL = 1e-6  # BASELINE.json north_star


@pytest.mark.parametrize("policy", ["FP64FP32", "FP64FP64", "FP64FP16"])
def test_fast_kbc_within_tolerance_and_exact_option_is_bit_exact(policy):
    vs, pp = init_hip("D3Q27", policy)
    ctx = get_context()
    lat = orc.Lattice("D3Q27")
    shape, omega, steps = (12, 10, 16), 1.9, 12
    f_np = orc.perturbed_init(shape, lat, policy, seed=3, amp_rho=0.02, amp_u=0.03)
    bm, mm = orc.build_masks(shape, lat, [])
    exp = orc.run(f_np, bm, mm, [], omega, lat, steps, policy, "KBC")
    outs = {}
    try:
        for exact in (0, 1):
            ctx.set_option("exact_math", exact)
            grid = grid_factory(shape)
            stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[], collision_type="KBC")
            f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
            f_0.assign(f_np)
            f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
            outs[exact] = f_0.numpy()
    finally:
        ctx.set_option("exact_math", 0)
    assert np.array_equal(outs[1], exp)
    err = float(np.abs(outs[0].astype(np.float64) - exp.astype(np.float64)).max())
    tol = {"FP64FP32": TOL, "FP64FP64": 1e-9, "FP64FP16": 2e-3}[policy]
    print(f"fast fp64 KBC {policy}: max |f - oracle| = {err:.3e}, bit-identical cells {np.mean(outs[0] == exp):.4f}")
    assert err <= tol
    # the two builds really are different code paths in fp64 storage (rounding-level differences are expected)
    if policy == "FP64FP64":
        assert err > 0.0


def test_fast_kbc_300_steps_at_omega_1p9_within_tolerance():
    """VERDICT r02 item 4's gate for the fp32 gamma reduction (cell.hpp COLL_G32: with a store type narrower than fp64 the two
    scalar products behind gamma run in fp32): BASELINE configs[4]'s set-up — D3Q27 KBC, fp64 compute / fp32 store, omega = 1.9 —
    on 24^3 for 300 steps stays within the north-star tolerance of the oracle (kbc.py:58-94)."""
    vs, pp = init_hip("D3Q27", "FP64FP32")
    lat = orc.Lattice("D3Q27")
    shape, omega, steps = (24, 24, 24), 1.9, 300
    f_np = orc.perturbed_init(shape, lat, "FP64FP32", seed=29, amp_rho=0.02, amp_u=0.03)
    bm, mm = orc.build_masks(shape, lat, [])
    exp = orc.run(f_np, bm, mm, [], omega, lat, steps, "FP64FP32", "KBC")
    grid = grid_factory(shape)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[], collision_type="KBC")
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_0.assign(f_np)
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    got = f_0.numpy()
    err = float(np.abs(got.astype(np.float64) - exp.astype(np.float64)).max())
    print(f"fast fp64 KBC (fp32 gamma reduction), 300 steps: max |f - oracle| = {err:.3e}, bit-identical cells {np.mean(got == exp):.4f}")
    assert err <= TOL
    assert float(np.abs(exp - f_np).max()) > 1e-4  # the state really evolved


def test_fast_kbc_golden_config5_twin():
    """The committed 12^3 golden vector of configs[4] (D3Q27 KBC FP64FP32) at the default (fast) setting."""
    g = golden("d3q27_kbc_12")
    vs, pp = init_hip("D3Q27", "FP64FP32")
    lat = orc.Lattice("D3Q27")
    shape = (12, 12, 12)
    grid = grid_factory(shape)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[], collision_type="KBC")
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_0.assign(orc.perturbed_init(shape, lat, "FP64FP32", seed=0))
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, float(g["omega"]), int(g["steps"]))
    assert np.abs(f_0.numpy().astype(np.float64) - g["f_FP64FP32"]).max() <= TOL


def test_fast_kbc_cavity_with_walls():
    """Halfway-wall cavity, D3Q27 KBC FP64FP32: boundary cells go through the same fast collision."""
    shape, omega, steps = (10, 12, 14), 1.7, 15
    grid, bcs, lat, obcs = hip_cavity_3d(shape, HalfwayBounceBackBC, lattice="D3Q27", policy="FP64FP32", u_lid=0.05)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type="KBC")
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    exp = orc.run(orc.initialize_eq(shape, lat, "FP64FP32"), o_bm, o_mm, obcs, omega, lat, steps, "FP64FP32", "KBC")
    assert np.abs(f_0.numpy().astype(np.float64) - exp.astype(np.float64)).max() <= TOL


@pytest.mark.parametrize("lattice,shape", [("D3Q19", (6, 16, 64)), ("D3Q27", (5, 8, 64))])
def test_fast_bgk_body_of_the_two_step_kernel_is_within_tolerance(lattice, shape):
    """Opt-in `fast_bgk=1` (cell.hpp: bgk_fast — one reciprocal, FMAs, shared E / O per pair) in the two-step kernel: rounding-
    level differences from the oracle; the default stays bit-exact."""
    vs, pp = init_hip(lattice)
    ctx = get_context()
    lat = orc.Lattice(lattice)
    f_np = orc.perturbed_init(shape, lat, seed=9, amp_rho=0.02, amp_u=0.03)
    bm, mm = orc.build_masks(shape, lat, [])
    steps = 10
    exp = orc.run(f_np, bm, mm, [], 1.6, lat, steps)
    outs = {}
    try:
        ctx.set_option("fuse2", 2)
        for fast in (0, 1):
            ctx.set_option("fast_bgk", fast)
            grid = grid_factory(shape)
            stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[])
            f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
            assert stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
            f_0.assign(f_np)
            f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.6, steps)
            outs[fast] = f_0.numpy()
    finally:
        ctx.set_option("fast_bgk", 0)
        ctx.set_option("fuse2", 1)
    assert np.array_equal(outs[0], exp)
    err = float(np.abs(outs[1].astype(np.float64) - exp.astype(np.float64)).max())
    print(f"fast BGK two-step {lattice}: max |f - oracle| = {err:.3e}")
    assert 0.0 < err <= TOL
