#!/usr/bin/env python3
"""Streaming-copy yardstick on this device: a plain copy kernel (4 and 16 bytes per lane) and
hipMemcpyAsync D2D over a buffer as large as one D3Q19 512^3 population field (10.2 GB).
Also the known-byte-count calibration run for the FETCH_SIZE / WRITE_SIZE counters."""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import xlb_amd  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    pp = PrecisionPolicy.FP32FP32
    vs = xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP)
    xlb_amd.init(vs, ComputeBackend.HIP, pp)
    ctx = get_context()
    grid = grid_factory((n, n, n))
    a = grid.create_field(19, fill_value=1.0)
    b = grid.create_field(19)
    info = a.info()
    nbytes = info["plane_stride"] * 19 * 4
    for label, fn in (("kernel 4 B/lane", lambda: b.copy_kernel_from(a, 4)), ("kernel 16 B/lane", lambda: b.copy_kernel_from(a, 16)),
                      ("hipMemcpyAsync D2D", lambda: b.copy_from(a))):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"{label:22s} {nbytes / 1e9:7.2f} GB each way  {dt * 1e3:8.3f} ms  {2 * nbytes / dt / 1e9:8.1f} GB/s (read+write)  {2 * nbytes / dt / 8e12:6.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
