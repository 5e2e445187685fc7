#!/usr/bin/env python3
"""Lid-driven cavity on several GPUs of one node — the multi-device counterpart of the reference's
examples/cfd/lid_driven_cavity_2d_distributed.py: one process per GPU, the domain cut into x-slabs, ghost planes exchanged over RCCL inside
the native step (once per pair of steps where the two-step kernel runs).  Boundary-condition indices stay GLOBAL; every rank builds the
masks of its own slab.

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 examples/cavity_3d_distributed_hip.py [--nx 1024 --ny 256 --nz 256]

(any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT works; without one the script runs on one GPU).  If the
RCCL communicator cannot be built — e.g. several ranks pointed at one GPU with XLB_HIP_DEVICE=0 — every rank falls back to the host-staged
transport and says so.
"""

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import xlb_amd as xlb
from xlb_amd import ComputeBackend, PrecisionPolicy, distribute
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import EquilibriumBC, HalfwayBounceBackBC
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper
from xlb_amd.precision_policy import Precision

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=512)
ap.add_argument("--ny", type=int, default=128)
ap.add_argument("--nz", type=int, default=128)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--re", type=float, default=400.0)
ap.add_argument("--u-lid", type=float, default=0.05)
args = ap.parse_args()

# halfway walls on both x faces: nothing is ever pulled across them, so the ring of slabs is a chain
rank, world = distribute.init_process_group(periodic_x=False, transport="rccl_or_host")

policy = PrecisionPolicy.FP32FP32
lattice = xlb.velocity_set.D3Q19(precision_policy=policy, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=lattice, default_backend=ComputeBackend.HIP, default_precision_policy=policy)

shape = (args.nx, args.ny, args.nz)
grid = grid_factory(shape)  # this rank's x-slab (+ two ghost planes per side when world > 1)
box = grid.bounding_box_indices(as_numpy=True)
box_no_edge = grid.bounding_box_indices(remove_edges=True, as_numpy=True)
walls = np.unique(np.concatenate([box[f] for f in ("bottom", "left", "right", "front", "back")], axis=1), axis=1)
bcs = [EquilibriumBC(rho=1.0, u=(args.u_lid, 0.0, 0.0), indices=box_no_edge["top"]), HalfwayBounceBackBC(indices=walls)]
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

visc = args.u_lid * (args.nz - 1) / args.re
omega = 1.0 / (3.0 * visc + 0.5)
if rank == 0:
    print(f"global {shape} on {world} rank(s) [{distribute.transport() or 'single GPU'}], slab of rank 0: {grid.local_shape}, omega {omega:.4f}")

distribute.barrier()
t0 = time.perf_counter()
f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, args.steps)
xlb.default_config.get_context().sync()
distribute.barrier()
dt = time.perf_counter() - t0

rho = grid.create_field(1, dtype=Precision.FP32)
u = grid.create_field(3, dtype=Precision.FP32)
Macroscopic()(f_0, rho, u)
u_all = distribute.gather_field(u)  # the global field on every rank (small domains; use per-slab output for large ones)
if rank == 0:
    print(f"{args.steps} steps in {dt:.2f} s: {np.prod(shape) * args.steps / dt / 1e6:.0f} MLUPS")
    mid = u_all[:, args.nx // 2, args.ny // 2, :]
    print(f"u_x along z through the centre: lid {mid[0, -2]:.4f}, centre {mid[0, args.nz // 2]:.4f}, floor {mid[0, 1]:.4f}; "
          f"max |u| {np.sqrt((u_all**2).sum(0)).max():.4f}")
    assert np.isfinite(u_all).all() and mid[0, -2] > 0.0
distribute.shutdown()
