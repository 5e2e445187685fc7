#!/usr/bin/env python3
"""The reference's 3-D lid-driven cavity benchmark loop (examples/performance/mlups_3d.py:193-242) on the HIP backend,
UNCHANGED in structure: construct the BCs and the stepper, then `f_0, f_1 = stepper(f_0, f_1, ...)` + swap per step, with a
device synchronisation before and after the timed loop.  The backend pairs consecutive calls into two-steps-per-pass kernel
launches behind the scenes (xlb_amd/operator/stepper/nse_stepper.py), so this loop runs at the speed of `stepper.run`.

    python examples/cavity_3d_reference_loop_hip.py [n=256] [steps=200]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xlb_amd  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC  # noqa: E402
from xlb_amd.operator.macroscopic import Macroscopic  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
precision_policy = PrecisionPolicy.FP32FP32
velocity_set = xlb_amd.velocity_set.D3Q19(precision_policy=precision_policy, compute_backend=ComputeBackend.HIP)
xlb_amd.init(velocity_set=velocity_set, default_backend=ComputeBackend.HIP, default_precision_policy=precision_policy)
grid = grid_factory((n, n, n))

# mlups_3d.py:193-204: lid = top face without its edges, walls = the other five faces (fullway bounce-back)
box = grid.bounding_box_indices(as_numpy=True)
box_no_edge = grid.bounding_box_indices(remove_edges=True, as_numpy=True)
lid = box_no_edge["top"]
walls = np.unique(np.concatenate([box[f] for f in ("bottom", "left", "right", "front", "back")], axis=1), axis=-1)
bcs = [EquilibriumBC(rho=1.0, u=(0.02, 0.0, 0.0), indices=lid), FullwayBounceBackBC(indices=walls)]
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type="BGK")
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
omega = 1.0

ctx = get_context()
for i in range(10):  # warm-up (mlups_3d.py:225-230)
    f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, i)
    f_0, f_1 = f_1, f_0
ctx.sync()
t0 = time.perf_counter()
for i in range(steps):  # mlups_3d.py:236-239
    f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, i)
    f_0, f_1 = f_1, f_0
ctx.sync()
dt = time.perf_counter() - t0
print(f"{n}^3, {steps} steps: {dt / steps * 1e3:.3f} ms/step, {n**3 * steps / dt / 1e6:.0f} MLUPS "
      f"({stepper._n_fused_pairs} fused pairs, {stepper._n_materialised} materialisations)")
rho, u = Macroscopic()(f_0, grid.create_field(1), grid.create_field(3))
print("max |u| =", float(np.abs(u.numpy()).max()), " mass defect =", float(rho.numpy().astype(np.float64).mean() - 1.0))
