"""Smagorinsky LES BGK and exact-difference forcing on the HIP backend vs
the oracle (smagorinsky_les_bgk.py:44-60, forced_collision.py:44-50, exact_difference_force.py:61-83).
No reference test pins them: parity unpinned by the reference."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC
from xlb_amd.operator.collision import SmagorinskyLESBGK
from xlb_amd.operator.equilibrium import QuadraticEquilibrium
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import init_hip, max_ulp_diff

pytestmark = pytest.mark.gpu

CASES = [
    ("D3Q19", (10, 8, 12), "FP32FP32", "SmagorinskyLESBGK", None),
    ("D3Q19", (10, 8, 12), "FP32FP32", "BGK", (1e-5, 0.0, -2e-6)),
    ("D3Q19", (6, 6, 8), "FP64FP64", "SmagorinskyLESBGK", (2e-5, 0.0, 0.0)),
    ("D3Q27", (6, 6, 8), "FP32FP32", "KBC", (1e-5, 1e-6, 0.0)),
    ("D3Q27", (6, 6, 8), "FP64FP32", "SmagorinskyLESBGK", None),
    ("D2Q9", (16, 12), "FP32FP32", "SmagorinskyLESBGK", (1e-5, 0.0)),
    ("D2Q9", (16, 12), "FP32FP32", "KBC", (0.0, 1e-5)),
]


@pytest.mark.parametrize("lattice,shape,policy,collision,force", CASES)
def test_channel_with_walls_vs_oracle(lattice, shape, policy, collision, force):
    """Periodic channel between two halfway walls (the turbulent-channel set-up in miniature)."""
    vs, pp = init_hip(lattice, policy)
    lat = orc.Lattice(lattice)
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    walls = [box["bottom"][i] + box["top"][i] for i in range(lat.d)]
    bc = HalfwayBounceBackBC(indices=walls)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc], collision_type=collision,
                                                force_vector=None if force is None else np.array(force))
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    obcs = [orc.BC(orc.KIND_HALFWAY_BB, bc.id, walls)]
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    f_np = orc.perturbed_init(shape, lat, policy, seed=17, amp_rho=0.01, amp_u=0.04)
    f_0.assign(f_np)
    steps = 8
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.8, steps)
    exp = orc.run(f_np, o_bm, o_mm, obcs, 1.8, lat, steps, policy, collision, force)
    out = f_0.numpy()
    tol = 1e-6 if policy != "FP64FP64" else 1e-12
    assert np.abs(out.astype(np.float64) - exp.astype(np.float64)).max() <= tol
    assert np.array_equal(out, exp), f"not bit-exact: max ulp {max_ulp_diff(out, exp)}"


@pytest.mark.parametrize("lattice,shape", [("D2Q9", (20, 14)), ("D3Q19", (8, 8, 8)), ("D3Q27", (6, 6, 6))])
def test_smagorinsky_standalone_operator(lattice, shape):
    vs, pp = init_hip(lattice)
    lat = orc.Lattice(lattice)
    grid = grid_factory(shape)
    f_np = orc.perturbed_init(shape, lat, seed=3, amp_u=0.05)
    f_np = (f_np * (1 + 0.02 * np.random.default_rng(1).standard_normal(f_np.shape))).astype(np.float32)
    f = grid.create_field(vs.q).assign(f_np)
    rho = grid.create_field(1)
    u = grid.create_field(vs.d)
    Macroscopic()(f, rho, u)
    feq = QuadraticEquilibrium()(rho, u, grid.create_field(vs.q))
    out = SmagorinskyLESBGK(smagorinsky_coef=0.2)(f, feq, grid.create_field(vs.q), 1.9).numpy()
    exp = orc.smagorinsky_les_bgk(f_np, feq.numpy(), 1.9, lat, 0.2)
    assert np.array_equal(out, exp), max_ulp_diff(out, exp)


def test_momentum_grows_with_the_force():
    vs, pp = init_hip("D3Q19")
    shape = (8, 8, 8)
    grid = grid_factory(shape)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[], force_vector=np.array([1e-4, 0.0, 0.0]))
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.0, 10)
    rho, u = orc.macroscopic(f_0.numpy(), orc.Lattice("D3Q19"))
    assert np.allclose(u[0], 10 * 1e-4, rtol=1e-3) and np.abs(u[1]).max() < 1e-7
