#!/usr/bin/env python3
"""Body-force driven channel on the HIP backend — the set-up of the reference's examples/cfd/turbulent_channel_3d.py: D3Q27 / KBC with an
exact-difference body force, no-slip walls as RegularizedBC("velocity", 0) on the two z faces, periodic in x and y, a perturbed log-law
initial field.  Prints the friction velocity from the mean profile against the target.

    python examples/turbulent_channel_3d_hip.py [--h 32] [--re-tau 180] [--steps 4000]
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
from xlb_amd.operator.boundary_condition import RegularizedBC
from xlb_amd.operator.equilibrium import QuadraticEquilibrium
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
from xlb_amd.precision_policy import Precision

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=32, help="channel half width in cells; the box is 6h x 3h x 2h (turbulent_channel_3d.py:60-66)")
ap.add_argument("--re-tau", type=float, default=180.0)
ap.add_argument("--u-tau", type=float, default=0.004)
ap.add_argument("--steps", type=int, default=4000)
args = ap.parse_args()

policy = PrecisionPolicy.FP32FP32
lattice = xlb.velocity_set.D3Q27(precision_policy=policy, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=lattice, default_backend=ComputeBackend.HIP, default_precision_policy=policy)

h = args.h
shape = (6 * h, 3 * h, 2 * h)
grid = grid_factory(shape)
visc = args.u_tau * h / args.re_tau
omega = 1.0 / (3.0 * visc + 0.5)
force = (args.re_tau * visc) ** 2 / h**3  # the pressure gradient that balances a wall stress rho u_tau^2 (turbulent_channel_3d.py:29-31)

box = grid.bounding_box_indices()
walls = [box["bottom"][i] + box["top"][i] for i in range(3)]
bc_walls = RegularizedBC("velocity", prescribed_value=(0.0, 0.0, 0.0), indices=walls)
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc_walls], collision_type="KBC", force_vector=(force, 0.0, 0.0))
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

# initial field: log-law mean + a divergence-free-ish sinusoidal perturbation that trips the transition, as an equilibrium
x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
zplus = np.minimum(z + 0.5, 2 * h - 0.5 - z) * args.u_tau / visc
u0 = np.zeros((3,) + shape)
u0[0] = args.u_tau * (np.log(np.maximum(zplus, 1.0)) / 0.41 + 5.5)
amp = 0.1 * u0[0].max()
u0[0] += amp * np.cos(2 * np.pi * y / shape[1] * 3) * np.sin(np.pi * z / shape[2])
u0[1] += amp * np.sin(2 * np.pi * x / shape[0] * 4) * np.sin(np.pi * z / shape[2])
u0[2] += 0.5 * amp * np.sin(2 * np.pi * x / shape[0] * 4) * np.cos(2 * np.pi * y / shape[1] * 3) * np.sin(np.pi * z / shape[2]) ** 2
rho0 = grid.create_field(1, dtype=Precision.FP32, fill_value=1.0)
u_init = grid.create_field(3, dtype=Precision.FP32)
u_init.assign(u0.astype(np.float32))
f_0 = QuadraticEquilibrium()(rho0, u_init, f_0)
print(f"grid {shape}, Re_tau {args.re_tau}, u_tau {args.u_tau}, viscosity {visc:.3e}, omega {omega:.4f}, body force {force:.3e}")

rho = grid.create_field(1, dtype=Precision.FP32)
u = grid.create_field(3, dtype=Precision.FP32)
ctx = xlb.default_config.get_context()
t0 = time.perf_counter()
done = 0
while done < args.steps:
    n = min(1000, args.steps - done)
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, n, first_timestep=done)
    done += n
    Macroscopic()(f_0, rho, u)
    un = u.numpy()
    mean = un[0].mean(axis=(0, 1))  # <u_x>(z)
    u_tau_now = np.sqrt(visc * abs(mean[1] - mean[0]) / 1.0)  # wall shear from the first two cell centres
    print(f"step {done}: bulk velocity {mean.mean():.5f}, centre-line {mean[h]:.5f}, u_tau from the wall gradient {u_tau_now:.5f} (target {args.u_tau})")
ctx.sync()
dt = time.perf_counter() - t0
print(f"{args.steps} steps in {dt:.2f} s: {np.prod(shape) * args.steps / dt / 1e6:.0f} MLUPS (incl. the statistics)")
assert np.isfinite(un).all() and mean[h] > 0.0
