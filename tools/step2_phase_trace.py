#!/usr/bin/env python3
"""Where the waves of the two-step kernel spend a plane: shader-clock stamps from a -DXLB_STEP2_TRACE build
(XLBHIP_LIB=<that build> python tools/step2_phase_trace.py [workload] [size]).  One block, six steady-state planes,
every wave; prints per wave the mean cycles between consecutive stamps:
  0 loop top | 1 phase B computed (LDS pulls + collision) | 2 stores issued | 3 barrier passed | (4 pulls arrived: TRACE=2 only)
  5 phase A collided + written to LDS | 6 next pulls issued | 7 barrier passed"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import xlb_amd  # noqa: E402
from bench import cavity_bcs  # noqa: E402
from xlb_amd import ComputeBackend, PrecisionPolicy, _lib  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.operator.boundary_condition import EquilibriumBC, HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "periodic"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    pp = PrecisionPolicy.FP32FP32
    vs = xlb_amd.velocity_set.D3Q19(pp, ComputeBackend.HIP)
    xlb_amd.init(vs, ComputeBackend.HIP, pp)
    ctx = get_context()
    ctx.set_option("fuse2", 2)
    for kv in filter(None, os.environ.get("XLB_TRACE_OPTS", "").split(",")):  # e.g. XLB_TRACE_OPTS=fuse2_strips=0
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    grid = grid_factory((n, n, n))
    bcs = [] if workload == "periodic" else cavity_bcs(grid, HalfwayBounceBackBC, EquilibriumBC)
    st = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f0, f1, bm, mm = st.prepare_fields()
    st.run(f0, f1, bm, mm, 1.0, 6)
    ctx.sync()
    lib = _lib.load()
    P, W, E = 6, 11, 8
    buf = (C.c_ulonglong * (P * W * E))()
    lib.xlbhip_debug_step2_trace.argtypes = [C.c_void_p, C.c_int]
    rc = lib.xlbhip_debug_step2_trace(buf, P * W * E)
    assert rc == 0, rc
    t = np.array(buf, dtype=np.int64).reshape(P, W, E)
    events = [e for e in range(E) if (t[:, :, e] != 0).all()]
    print(f"# {workload} {n}^3: cycles between stamps (mean over {P} planes), events present: {events}")
    print("wave " + " ".join(f"{a}->{b:>1d}".rjust(8) for a, b in zip(events[:-1], events[1:])) + "  7->0(next)".rjust(12) + "   plane".rjust(10))
    for w in range(W):
        seg = [float(np.mean(t[:, w, b] - t[:, w, a])) for a, b in zip(events[:-1], events[1:])]
        nxt = float(np.mean(t[1:, w, events[0]] - t[:-1, w, events[-1]]))
        plane = float(np.mean(t[1:, w, events[0]] - t[:-1, w, events[0]]))
        print(f"{w:4d} " + " ".join(f"{v:8.0f}" for v in seg) + f"{nxt:12.0f}{plane:10.0f}")


if __name__ == "__main__":
    main()
