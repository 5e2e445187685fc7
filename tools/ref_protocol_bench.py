#!/usr/bin/env python3
"""What an unmodified XLB driver sees: the reference's own loop (mlups_3d.py:225-242) — stepper(f_0, f_1, ...) + swap per step,
device sync before and after — against the native stepper.run() loop, cavity 512^3 (configs[2])."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xlb_amd  # noqa: E402
from bench import cavity_bcs  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.boundary_condition import EquilibriumBC, HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 200
pp = PrecisionPolicy.FP32FP32
vs = xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP)
xlb_amd.init(vs, ComputeBackend.HIP, pp)
ctx = get_context()
for lazy in (True, False):
    grid = grid_factory((n, n, n))
    st = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=cavity_bcs(grid, HalfwayBounceBackBC, EquilibriumBC), backend_config={"lazy_pairs": lazy})
    f_0, f_1, bm, mm = st.prepare_fields()
    for i in range(10):
        f_0, f_1 = st(f_0, f_1, bm, mm, 1.0, i)
        f_0, f_1 = f_1, f_0
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        f_0, f_1 = st(f_0, f_1, bm, mm, 1.0, i)
        f_0, f_1 = f_1, f_0
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"reference loop, lazy_pairs={lazy}: {dt / steps * 1e3:.3f} ms/step  {n**3 * steps / dt / 1e6:.0f} MLUPS", flush=True)
    if lazy:
        # (the last pair left f(t+1) virtual; the first use of that field materialises it through a temporary third field — 10 GB
        # allocated and freed.  Keep that one-off out of the native loop's timing.)
        f_0, f_1 = st.run(f_0, f_1, bm, mm, 1.0, 2)
        ctx.sync()
        t0 = time.perf_counter()
        f_0, f_1 = st.run(f_0, f_1, bm, mm, 1.0, steps)
        ctx.sync()
        dt = time.perf_counter() - t0
        print(f"stepper.run (native loop):       {dt / steps * 1e3:.3f} ms/step  {n**3 * steps / dt / 1e6:.0f} MLUPS", flush=True)
    for f in (f_0, f_1, bm, mm):
        f.free()
