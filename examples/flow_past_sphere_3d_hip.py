#!/usr/bin/env python3
"""Flow past a sphere on the HIP backend — the set-up of the reference's examples/cfd/flow_past_sphere_3d.py:
fullway walls, a Regularized velocity inlet with a parabolic PROFILE, an extrapolation outflow and a halfway
bounce-back sphere given by interior indices; Vorticity / QCriterion on the result.

    python examples/flow_past_sphere_3d_hip.py [--nx 256 --ny 96 --nz 96] [--steps 2000] [--re 100]
"""

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import xlb_amd as xlb
from xlb_amd import ComputeBackend, PrecisionPolicy
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import ExtrapolationOutflowBC, FullwayBounceBackBC, HalfwayBounceBackBC, RegularizedBC
from xlb_amd.operator.force import MomentumTransfer
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.postprocess import QCriterion, Vorticity
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
from xlb_amd.precision_policy import Precision

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=256)
ap.add_argument("--ny", type=int, default=96)
ap.add_argument("--nz", type=int, default=96)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--re", type=float, default=100.0)
ap.add_argument("--u-max", type=float, default=0.04)
ap.add_argument("--fuse2", type=int, default=1, help="0: single-step kernel only; 1: two steps per pass where the library finds it eligible")
args = ap.parse_args()

policy = PrecisionPolicy.FP32FP32
lattice = xlb.velocity_set.D3Q19(precision_policy=policy, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=lattice, default_backend=ComputeBackend.HIP, default_precision_policy=policy)

xlb.default_config.get_context().set_option("fuse2", args.fuse2)
shape = (args.nx, args.ny, args.nz)
grid = grid_factory(shape)
box = grid.bounding_box_indices()
box_no_edge = grid.bounding_box_indices(remove_edges=True)
walls = [box["bottom"][i] + box["top"][i] + box["front"][i] + box["back"][i] for i in range(3)]
walls = np.unique(np.array(walls), axis=-1).tolist()

radius = args.ny // 12
x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
sphere = [s.tolist() for s in np.where((x - args.nx // 6) ** 2 + (y - args.ny // 2) ** 2 + (z - args.nz // 2) ** 2 < radius**2)]


def bc_profile():
    """Parabolic inlet, (3, ny, nz): broadcast along x like the reference's bc_profile_jax (flow_past_sphere_3d.py:64-81)."""
    yy, zz = np.meshgrid(np.arange(args.ny), np.arange(args.nz), indexing="ij")
    hy, hz = args.ny - 1.0, args.nz - 1.0
    r2 = (2.0 * (yy - hy / 2.0) / hy) ** 2 + (2.0 * (zz - hz / 2.0) / hz) ** 2
    ux = args.u_max * np.maximum(0.0, 1.0 - r2)
    return np.stack([ux, np.zeros_like(ux), np.zeros_like(ux)])


bc_left = RegularizedBC("velocity", profile=bc_profile, indices=box_no_edge["left"])
bc_walls = FullwayBounceBackBC(indices=walls)
bc_outlet = ExtrapolationOutflowBC(indices=box_no_edge["right"])
bc_sphere = HalfwayBounceBackBC(indices=sphere)
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc_walls, bc_left, bc_outlet, bc_sphere], collision_type="BGK")
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

visc = args.u_max * (2 * radius) / args.re
omega = 1.0 / (3.0 * visc + 0.5)
print(f"grid {shape}, sphere radius {radius}, Re {args.re}, omega {omega:.4f}")

t0 = time.perf_counter()
f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, args.steps)
xlb.default_config.get_context().sync()
dt = time.perf_counter() - t0
print(f"{args.steps} steps in {dt:.2f} s: {np.prod(shape) * args.steps / dt / 1e6:.0f} MLUPS")

rho = grid.create_field(1, dtype=Precision.FP32)
u = grid.create_field(3, dtype=Precision.FP32)
Macroscopic()(f_0, rho, u)
vort, mag = Vorticity()(u, bc_mask, grid.create_field(3, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
_, q = QCriterion()(u, bc_mask, grid.create_field(1, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
force = MomentumTransfer(bc_sphere)(f_0, f_1, bc_mask, missing_mask)
cd = 2.0 * force[0] / (1.0 * (2.0 / 3.0 * args.u_max) ** 2 * np.pi * radius**2)  # mean of the parabolic profile ~ 2/3 u_max ... rough
print(f"force on the sphere {force}, drag coefficient (rough, channel-confined) {cd:.2f}")
un, rn, mn, qn = u.numpy(), rho.numpy(), mag.numpy(), q.numpy()
fluid = bc_mask.numpy()[0] == 0
print(f"rho in [{rn[0][fluid].min():.4f}, {rn[0][fluid].max():.4f}], max |u| {np.sqrt((un**2).sum(0))[fluid].max():.4f}, "
      f"wake u_x behind the sphere {un[0, args.nx // 6 + 2 * radius, args.ny // 2, args.nz // 2]:.4f}, "
      f"max |vorticity| {mn.max():.4f}, max Q {qn.max():.3e}")
assert np.isfinite(un).all() and un[0][fluid].max() > 0.5 * args.u_max
