#!/usr/bin/env python3
"""Does the streaming-copy yardstick (xlbhip_field_copy_kernel) really copy a field of more than 2^32 16-byte words?
(BENCH r02's copy_yardstick_gbs at 4096 x 512 x 512 was 40 546 GB/s: some of the field was skipped.)  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import xlb_amd  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402

pp = PrecisionPolicy.FP32FP32
vs = xlb_amd.velocity_set.D3Q19(precision_policy=pp, compute_backend=ComputeBackend.HIP)
xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
ctx = get_context()
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
grid = grid_factory((nx, 512, 512))
a = grid.create_field(19, fill_value=1.5)
b = grid.create_field(19, fill_value=0.0)
ctx.sync()
b.copy_kernel_from(a, 16)
ctx.sync()
t0 = time.perf_counter()
for _ in range(3):
    b.copy_kernel_from(a, 16)
ctx.sync()
dt = (time.perf_counter() - t0) / 3
info = a.info()
gb = 2 * info["plane_stride"] * 19 * 4 / 1e9
bad = [(l, x) for l in (0, 9, 18) for x in (0, nx // 2, nx - 1) if not np.all(b.get_plane(l, x) == np.float32(1.5))]
print(f"copy of {gb / 2:.1f} GB: {dt * 1e3:.2f} ms -> {gb / dt:.0f} GB/s; planes not copied: {bad}")
