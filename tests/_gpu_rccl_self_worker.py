"""One rank, REAL RCCL communicator (ncclCommInitRank with n_ranks = 1): the ghost planes are refilled
by ncclSend/ncclRecv to self inside ncclGroupStart/End on the communication stream, overlapped with
the interior kernel — the exact code path of the multi-GPU runs, minus the wire."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import xlb_numpy as orc  # noqa: E402
from xlb_amd import _lib  # noqa: E402
from xlb_amd.default_config import get_context  # noqa: E402
from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC  # noqa: E402
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper  # noqa: E402
from _util import hip_cavity_3d, init_hip  # noqa: E402


def main():
    init_hip("D3Q19")
    ctx = get_context()
    ctx.comm_init(0, 1, _lib.comm_unique_id())
    shape = (12, 10, 16)
    grid, bcs, lat, obcs = hip_cavity_3d(shape, HalfwayBounceBackBC, backend_config={"halo": True})
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_np = orc.perturbed_init(shape, lat, seed=8)
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    ok = True
    for overlap in (1, 0):
        ctx.set_option("overlap", overlap)
        f_0.assign(f_np)
        a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.5, 9)
        exp = orc.run(f_np, o_bm, o_mm, obcs, 1.5, lat, 9)
        ok &= bool(np.array_equal(a.numpy(), exp))
    print("RCCL_SELF_OK" if ok else "RCCL_SELF_MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
