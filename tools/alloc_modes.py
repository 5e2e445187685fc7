#!/usr/bin/env python3
"""Is the two-step kernel's run-to-run spread (2.35 / 2.50 ms per step on ONE box, bimodal per process) a property of where the
fields were allocated?  Re-creates the fields several times in one process and times 40 steps + the copy yardstick each time,
printing the device addresses.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import xlb_amd  # noqa: E402
from bench import cavity_bcs  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.boundary_condition import EquilibriumBC, HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402

pp = PrecisionPolicy.FP32FP32
vs = xlb_amd.velocity_set.D3Q19(precision_policy=pp, compute_backend=ComputeBackend.HIP)
xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
ctx = get_context()
n = 512
grid = grid_factory((n, n, n))
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    st = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=cavity_bcs(grid, HalfwayBounceBackBC, EquilibriumBC))
    f0, f1, bm, mm = st.prepare_fields()
    a0, a1 = f0.info()["device_ptr"], f1.info()["device_ptr"]
    st.run(f0, f1, bm, mm, 1.0, 10)
    ctx.sync()
    (f0, f1), ms = st.run_timed(f0, f1, bm, mm, 1.0, 40)
    f1.copy_kernel_from(f0, 16)
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(5):
        f1.copy_kernel_from(f0, 16)
    ctx.sync()
    info = f0.info()
    gbs = 2 * info["plane_stride"] * 19 * 4 * 5 / (time.perf_counter() - t1) / 1e9
    print(f"trial {trial}: f_0 @ {a0:#x} f_1 @ {a1:#x} (delta {(a1 - a0) / 2**20:.1f} MiB)  {ms / 40:.4f} ms/step  copy {gbs:.0f} GB/s", flush=True)
    if trial % 2 == 1:
        keep.append((f0, f1))  # (hold some allocations so that the next ones land elsewhere)
    else:
        for f in (f0, f1, bm, mm):
            f.free()
