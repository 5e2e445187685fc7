#!/usr/bin/env python3
"""Flow past a ROTATING sphere on the HIP backend — the set-up of the reference's examples/cfd/rotating_sphere_3d.py: D3Q27 / KBC, a
Regularized velocity inlet, a do-nothing outlet, periodic side faces, and the sphere as a triangle mesh behind a
HybridBC("nonequilibrium_regularized") with RAY voxelisation, wall distances and a wall-velocity PROFILE omega x (r - centre); drag and
lift through MomentumTransfer.

    python examples/rotating_sphere_3d_hip.py [--diam 16] [--steps 2000] [--re 200] [--spin -0.2]
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
from xlb_amd.operator.boundary_condition import DoNothingBC, HybridBC, RegularizedBC
from xlb_amd.operator.boundary_masker import MeshVoxelizationMethod
from xlb_amd.operator.force import MomentumTransfer
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
from xlb_amd.precision_policy import Precision


def icosphere(centre, radius, subdivisions):
    """Triangles (n, 3, 3) of a subdivided icosahedron projected onto the sphere (stands in for the STL the reference loads)."""
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1],
                  [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    tri = v[f]
    for _ in range(subdivisions):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    tri = tri / np.linalg.norm(tri, axis=2, keepdims=True)
    return (np.asarray(centre) + radius * tri).reshape(-1, 3).astype(np.float32)


ap = argparse.ArgumentParser()
ap.add_argument("--diam", type=int, default=16, help="sphere diameter in cells; the box is 10 x 7 x 7 diameters (rotating_sphere_3d.py:41)")
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--re", type=float, default=200.0)
ap.add_argument("--wind", type=float, default=0.04)
ap.add_argument("--spin", type=float, default=-0.2, help="non-dimensional rotation rate omega D / (2 U)")
args = ap.parse_args()

policy = PrecisionPolicy.FP32FP32
lattice = xlb.velocity_set.D3Q27(precision_policy=policy, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=lattice, default_backend=ComputeBackend.HIP, default_precision_policy=policy)

d = args.diam
shape = (10 * d, 7 * d, 7 * d)
grid = grid_factory(shape)
box = grid.bounding_box_indices()
box_no_edge = grid.bounding_box_indices(remove_edges=True)
centre = np.array([shape[0] / 3.0 + d / 2.0, shape[1] / 2.0, shape[2] / 2.0])
rot_rate = 2.0 * args.wind * args.spin / d
spin_axis = np.array([0.0, rot_rate, 0.0])


def wall_velocity(cells):
    """omega x (r - centre) at the sphere's boundary cells ((3, n) indices -> (3, n) velocities)."""
    return np.cross(spin_axis.reshape(1, 3), (cells.astype(np.float64) - centre.reshape(3, 1)).T).T


bc_inlet = RegularizedBC("velocity", prescribed_value=(args.wind, 0.0, 0.0), indices=box_no_edge["left"])
bc_outlet = DoNothingBC(indices=box["right"])
bc_sphere = HybridBC("nonequilibrium_regularized", profile=wall_velocity, mesh_vertices=icosphere(centre, d / 2.0, 3),
                     voxelization_method=MeshVoxelizationMethod("RAY"), use_mesh_distance=True)
# (no BC on the side faces: they stay periodic, as in the reference's driver)
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc_inlet, bc_outlet, bc_sphere], collision_type="KBC")
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

visc = args.wind * d / args.re
omega = 1.0 / (3.0 * visc + 0.5)
print(f"grid {shape}, sphere diameter {d}, Re {args.re}, spin {args.spin}, omega {omega:.4f}, "
      f"{int((bc_mask.numpy()[0] == bc_sphere.id).sum())} boundary cells on the sphere")

momentum_transfer = MomentumTransfer(bc_sphere)
area = np.pi * d**2 / 4.0
ctx = xlb.default_config.get_context()
t0 = time.perf_counter()
done = 0
while done < args.steps:
    n = min(500, args.steps - done)
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, n, first_timestep=done)
    done += n
    force = momentum_transfer(f_0, f_1, bc_mask, missing_mask)
    cd, cl = 2.0 * force[0] / (args.wind**2 * area), 2.0 * force[2] / (args.wind**2 * area)
    print(f"step {done}: drag coefficient {cd:.4f}, lift coefficient {cl:.4f}")
ctx.sync()
dt = time.perf_counter() - t0
print(f"{args.steps} steps in {dt:.2f} s: {np.prod(shape) * args.steps / dt / 1e6:.0f} MLUPS (incl. the force evaluations)")

rho = grid.create_field(1, dtype=Precision.FP32)
u = grid.create_field(3, dtype=Precision.FP32)
Macroscopic()(f_0, rho, u)
un = u.numpy()
fluid = bc_mask.numpy()[0] == 0
speed = np.sqrt((un**2).sum(0))
print(f"max |u| in the fluid {speed[fluid].max():.4f}; u_x one diameter behind the sphere {un[0, int(centre[0]) + d, shape[1] // 2, shape[2] // 2]:.4f}")
assert np.isfinite(un[:, fluid]).all() and cd > 0.0
