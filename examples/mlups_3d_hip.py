#!/usr/bin/env python3
"""The reference's performance harness (examples/performance/mlups_3d.py) on the HIP backend, with its command line: a lid-driven cavity
with fullway walls in a cube, the reference's own driver loop (stepper(f_0, f_1, ...) + swap, 10 warm-up steps, device sync around the
timed region), MLUPS per repetition and their statistics.

    python examples/mlups_3d_hip.py 512 200 hip fp32/fp32 [--velocity_set D3Q27] [--collision_model KBC] [--repetitions 3]
                                                        [--export_final_velocity] [--measure_scalability --gpu_devices [0,1,2,3]]

--measure_scalability runs the same fixed cube on 1, 2, 4, ... GPUs through bench.py's launcher (one process per GPU, slab
decomposition along x, halo over RCCL) and prints the speed-ups — the reference sweeps device counts the same way (mlups_3d.py:546-556).
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

PRECISIONS = {"fp32/fp32": "FP32FP32", "fp64/fp64": "FP64FP64", "fp64/fp32": "FP64FP32", "fp32/fp16": "FP32FP16", "fp64/fp16": "FP64FP16"}

ap = argparse.ArgumentParser(description="MLUPS of the lid-driven cavity on the HIP backend (the reference harness's command line)")
ap.add_argument("cube_edge", type=int)
ap.add_argument("num_steps", type=int)
ap.add_argument("compute_backend", choices=["hip"])
ap.add_argument("precision", choices=list(PRECISIONS))
ap.add_argument("--gpu_devices", default=None, help="e.g. [0,1,2,3]: the device counts --measure_scalability sweeps (default: 1)")
ap.add_argument("--velocity_set", default="D3Q19", choices=["D3Q19", "D3Q27"])
ap.add_argument("--collision_model", default="BGK", choices=["BGK", "KBC"])
ap.add_argument("--export_final_velocity", action="store_true", help="write the final velocity field as a VTK file")
ap.add_argument("--measure_scalability", action="store_true")
ap.add_argument("--repetitions", type=int, default=1)
args = ap.parse_args()
if args.collision_model == "KBC" and args.velocity_set != "D3Q27":
    ap.error("KBC requires D3Q27")

policy = PRECISIONS[args.precision]
n = args.cube_edge

if args.measure_scalability:
    devices = json.loads(args.gpu_devices) if args.gpu_devices else [0]
    counts = [c for c in (1, 2, 4, 8, 16) if c <= len(devices)]
    rows = []
    for c in counts:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(c), "--workload", "cavity_fullway", "--global-shape", f"{n}x{n}x{n}",
               "--steps", str(args.num_steps), "--lattice", args.velocity_set, "--collision", args.collision_model, "--policy", policy,
               "--cpu-baseline-seconds", "0"]
        out = subprocess.run(cmd, capture_output=True, text=True)
        if out.returncode != 0:
            sys.exit(f"{c} GPU(s): bench.py failed\n{out.stderr[-2000:]}")
        line = json.loads(out.stdout.strip().splitlines()[-1])
        rows.append((c, line["value"], line["config"]["decomposition"]))
    print(f"\nScalability, {n}^3 {args.velocity_set} {args.collision_model} {args.precision}, {args.num_steps} steps:")
    for c, mlups, how in rows:
        print(f"  {c:2d} GPU(s): {mlups:10.1f} MLUPS   x{mlups / rows[0][1]:5.2f}   ({how})")
    sys.exit(0)

import xlb_amd as xlb
from xlb_amd import ComputeBackend, PrecisionPolicy
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC
from xlb_amd.operator.macroscopic import Macroscopic
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

pp = PrecisionPolicy[policy]
vs = getattr(xlb.velocity_set, args.velocity_set)(precision_policy=pp, compute_backend=ComputeBackend.HIP)
xlb.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
ctx = xlb.default_config.get_context()
print(f"{n}^3 = {n**3:,} lattice points, {args.num_steps} steps x {args.repetitions} repetition(s), {args.velocity_set} {args.collision_model} {args.precision}")

grid = grid_factory((n, n, n))
box = grid.bounding_box_indices(as_numpy=True)
box_no_edge = grid.bounding_box_indices(remove_edges=True, as_numpy=True)
walls = np.concatenate([box[f] for f in ("bottom", "left", "right", "front", "back")], axis=1).astype(np.int64)
keys = np.unique((walls[0] * n + walls[1]) * n + walls[2])  # (np.unique(walls, axis=-1) of the reference driver, on linear keys)
walls = np.stack([keys // (n * n), (keys // n) % n, keys % n]).astype(np.int32)
bcs = [EquilibriumBC(rho=1.0, u=(0.02, 0.0, 0.0), indices=box_no_edge["top"]), FullwayBounceBackBC(indices=walls)]
stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs, collision_type=args.collision_model)
omega = 1.0
f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()

for i in range(10):
    f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, i)
    f_0, f_1 = f_1, f_0
ctx.sync()
elapsed = []
for _ in range(args.repetitions):
    t0 = time.time()
    for i in range(args.num_steps):
        f_0, f_1 = stepper(f_0, f_1, bc_mask, missing_mask, omega, i)
        f_0, f_1 = f_1, f_0
    ctx.sync()
    elapsed.append(time.time() - t0)

mlups = np.array([n**3 * args.num_steps / t / 1e6 for t in elapsed])
print(f"elapsed per repetition: {', '.join(f'{t:.3f} s' for t in elapsed)}")
print(f"MLUPS: mean {mlups.mean():.1f}, std {mlups.std():.1f}, min {mlups.min():.1f}, max {mlups.max():.1f}")
if args.export_final_velocity:
    from xlb_amd.precision_policy import Precision
    from xlb_amd.utils import save_fields_vtk

    rho = grid.create_field(1, dtype=Precision.FP32)
    u = grid.create_field(3, dtype=Precision.FP32)
    Macroscopic()(f_0, rho, u)
    un = u.numpy()
    save_fields_vtk({"u_x": un[0], "u_y": un[1], "u_z": un[2]}, timestep=10 + args.num_steps * args.repetitions, prefix=f"mlups_3d_size_{n}")
