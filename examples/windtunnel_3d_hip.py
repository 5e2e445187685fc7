#!/usr/bin/env python3
"""Wind tunnel around a triangle mesh on the HIP backend — the set-up of the reference's examples/cfd/windtunnel_3d.py: D3Q27 / KBC, a
Regularized velocity inlet, fullway tunnel walls, an extrapolation outflow, the body from an STL file behind a halfway bounce-back wall
(or, --hybrid, the curved-wall HybridBC with wall distances), drag and lift through MomentumTransfer, VTK and PNG output through
xlb_amd.utils.  The reference's driver loop (stepper(...) + swap) is used as is.

    python examples/windtunnel_3d_hip.py [--stl body.stl] [--nx 256] [--steps 3000] [--re 50000] [--hybrid] [--out DIR]

Without --stl a bluff body (an ellipsoid with a flat underside) is generated, written as STL and read back.
"""

import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import xlb_amd as xlb
from xlb_amd import ComputeBackend, PrecisionPolicy
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import ExtrapolationOutflowBC, FullwayBounceBackBC, HalfwayBounceBackBC, HybridBC, RegularizedBC
from xlb_amd.operator.boundary_masker import MeshVoxelizationMethod
from xlb_amd.operator.force import MomentumTransfer
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
from xlb_amd.precision_policy import Precision
from xlb_amd.utils import load_stl, save_fields_vtk, save_image, save_stl


def bluff_body(n_lat=24, n_lon=48):
    """Triangle soup of an ellipsoid (4 : 1.6 : 1.2) whose lower third is cut flat — closed, outward-facing."""
    th = np.linspace(0.0, np.pi, n_lat + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    t, p = np.meshgrid(th, ph, indexing="ij")
    pts = np.stack([2.0 * np.sin(t) * np.cos(p), 0.8 * np.sin(t) * np.sin(p), np.maximum(0.6 * np.cos(t), -0.2)], axis=-1)
    a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
    tri = np.concatenate([np.stack([a, b, c], axis=2).reshape(-1, 3, 3), np.stack([a, c, d], axis=2).reshape(-1, 3, 3)])
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    return tri[area > 1e-12].reshape(-1, 3)


ap = argparse.ArgumentParser()
ap.add_argument("--stl", default="")
ap.add_argument("--nx", type=int, default=256, help="tunnel length in cells; the tunnel is nx x nx/2 x nx/4, the body nx/4 long (windtunnel_3d.py:26-29, 87)")
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--re", type=float, default=50000.0)
ap.add_argument("--wind", type=float, default=0.02)
ap.add_argument("--hybrid", action="store_true", help="HybridBC('nonequilibrium_regularized') with wall distances instead of the halfway wall")
ap.add_argument("--out", default="", help="directory for the VTK / PNG output (none by default)")
ap.add_argument("--every", type=int, default=1000)
args = ap.parse_args()

policy = PrecisionPolicy.FP32FP32
lattice = xlb.velocity_set.D3Q27(precision_policy=policy, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=lattice, default_backend=ComputeBackend.HIP, default_precision_policy=policy)

shape = (args.nx, args.nx // 2, args.nx // 4)
grid = grid_factory(shape)
box = grid.bounding_box_indices()
box_no_edge = grid.bounding_box_indices(remove_edges=True)
walls = [box["bottom"][i] + box["top"][i] + box["front"][i] + box["back"][i] for i in range(3)]
walls = np.unique(np.array(walls), axis=-1).tolist()

stl = args.stl
if not stl:
    stl = os.path.join(tempfile.mkdtemp(prefix="xlb_amd_"), "bluff_body.stl")
    save_stl(stl, bluff_body())
verts = load_stl(stl).astype(np.float64)
verts -= verts.min(axis=0)
extents = verts.max(axis=0)
length = shape[0] / 4.0
dx = extents.max() / length  # physical size of a voxel: the body's largest extent spans nx / 4 voxels
method = MeshVoxelizationMethod("RAY")
clearance = 2.0  # the RAY voxeliser tags the fluid voxels next to the surface: keep them off the ground (windtunnel_3d.py:91-97)
body = verts / dx + np.array([shape[0] / 4.0, (shape[1] - extents[1] / dx) / 2.0, clearance])
cross_section = np.prod(extents[1:] / dx)

bc_inlet = RegularizedBC("velocity", prescribed_value=(args.wind, 0.0, 0.0), indices=box_no_edge["left"])
bc_walls = FullwayBounceBackBC(indices=walls)
bc_outlet = ExtrapolationOutflowBC(indices=box_no_edge["right"])
if args.hybrid:
    bc_body = HybridBC("nonequilibrium_regularized", mesh_vertices=body.astype(np.float32), voxelization_method=method, use_mesh_distance=True)
else:
    bc_body = HalfwayBounceBackBC(mesh_vertices=body.astype(np.float32), voxelization_method=method)
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc_walls, bc_inlet, bc_outlet, bc_body], collision_type="KBC")
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

visc = args.wind * length / args.re
omega = 1.0 / (3.0 * visc + 0.5)
print(f"tunnel {shape}, body {stl} ({len(verts) // 3} triangles, {length:.0f} cells long), Re {args.re}, omega {omega:.5f}, "
      f"{int((bc_mask.numpy()[0] == bc_body.id).sum())} boundary cells on the body")

momentum_transfer = MomentumTransfer(bc_body)
macro = Macroscopic()
rho = grid.create_field(1, dtype=Precision.FP32)
u = grid.create_field(3, dtype=Precision.FP32)
ctx = xlb.default_config.get_context()
t0 = time.perf_counter()
for step in range(args.steps):
    f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, step)
    f_0, f_1 = f_1, f_0
    if (step + 1) % args.every == 0 or step == args.steps - 1:
        force = momentum_transfer(f_0, f_1, bc_mask, missing_mask)
        cd, cl = 2.0 * force[0] / (args.wind**2 * cross_section), 2.0 * force[2] / (args.wind**2 * cross_section)
        print(f"step {step + 1}: drag coefficient {cd:.4f}, lift coefficient {cl:.4f}")
        if args.out:
            macro(f_0, rho, u)
            un = u.numpy()[:, 1:-1, 1:-1, 1:-1]
            fields = {"u_magnitude": np.sqrt((un**2).sum(0)), "u_x": un[0], "u_y": un[1], "u_z": un[2], "rho": rho.numpy()[0, 1:-1, 1:-1, 1:-1]}
            save_fields_vtk(fields, timestep=step + 1, output_dir=args.out)
            save_image(fields["u_magnitude"][:, shape[1] // 2 - 1, :], timestep=step + 1, prefix=os.path.join(args.out, "windtunnel"))
ctx.sync()
dt = time.perf_counter() - t0
print(f"{args.steps} steps in {dt:.2f} s: {np.prod(shape) * args.steps / dt / 1e6:.0f} MLUPS")
macro(f_0, rho, u)
un = u.numpy()
fluid = bc_mask.numpy()[0] == 0
assert np.isfinite(un[:, fluid]).all() and cd > 0.0
print(f"max |u| in the fluid {np.sqrt((un**2).sum(0))[fluid].max():.4f}")
