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
