#!/usr/bin/env python3
"""Times the whole-field operators (output pass etc.) at a given size: ms, GB/s of algorithmic bytes."""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import xlb_amd  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.collision import BGK  # noqa: E402
from xlb_amd.operator.equilibrium import QuadraticEquilibrium  # noqa: E402
from xlb_amd.operator.macroscopic import Macroscopic  # noqa: E402
from xlb_amd.operator.stream import Stream  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = 10
    pp = PrecisionPolicy.FP32FP32
    vs = xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP)
    xlb_amd.init(vs, ComputeBackend.HIP, pp)
    ctx = get_context()
    grid = grid_factory((n, n, n))
    f = grid.create_field(19, fill_value=0.05)
    g = grid.create_field(19)
    rho = grid.create_field(1, fill_value=1.0)
    u = grid.create_field(3)
    cells = float(n) ** 3
    ops = [
        ("Macroscopic (q+1+d)*4 B", lambda: Macroscopic()(f, rho, u), (19 + 4) * 4),
        ("QuadraticEquilibrium (1+d+q)*4 B", lambda: QuadraticEquilibrium()(rho, u, g), (19 + 4) * 4),
        ("Stream 2q*4 B", lambda: Stream()(f, g), 2 * 19 * 4),
        ("BGK 3q*4 B", lambda: BGK()(f, g, g, 1.0), 3 * 19 * 4),
    ]
    for name, fn, bytes_per_cell in ops:
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:36s} {dt * 1e3:8.3f} ms  {bytes_per_cell * cells / dt / 1e9:8.1f} GB/s  {bytes_per_cell * cells / dt / 8e12:6.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
