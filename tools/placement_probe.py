#!/usr/bin/env python3
"""Does the placement of the population fields in device memory change the two-step kernel's time?  Allocates a dummy
buffer of varying size before the fields (shifts their addresses) and times the periodic 512^3 box each time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import xlb_amd  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy, _lib  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402

pp = PrecisionPolicy.FP32FP32
vs = xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP)
xlb_amd.init(vs, ComputeBackend.HIP, pp)
ctx = get_context()
n = 512
for pad_kb in [0, 4, 64, 1024, 2048 + 4, 65536, 1024 * 1024 + 128]:
    dummy = _lib.Field(ctx, 1, (max(1, pad_kb), 16, 16), _lib.F32) if pad_kb else None  # pad_kb KiB
    grid = grid_factory((n, n, n))
    st = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[])
    f0, f1, bm, mm = st.prepare_fields()
    st.run(f0, f1, bm, mm, 1.0, 4)
    ctx.sync()
    ts = []
    for _ in range(3):
        _, ms = st.run_timed(f0, f1, bm, mm, 1.0, 20)
        ts.append(ms / 20)
    i0, i1 = f0.info(), f1.info()
    print(f"pad {pad_kb:8d} KiB  f0 @ {i0['device_ptr']:#x} f1 @ {i1['device_ptr']:#x} stride {i0['plane_stride']}  ms/step {min(ts):.4f} {np.median(ts):.4f}", flush=True)
    for f in (f0, f1, bm, mm):
        f.free()
    if dummy is not None:
        dummy.free()
