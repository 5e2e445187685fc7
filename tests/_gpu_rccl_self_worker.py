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
    # the all-reduce that decides "pairs or single steps" for all ranks (comm_all_min) had never executed —
    # with one rank it is skipped; this option sends it through ncclAllReduce on the one-rank communicator too
    ctx.set_option("comm_self_test", 1)
    ok = True
    # (12, 10, 16): single-step kernel with one or two ghost planes; (20, 16, 64): two-step kernel, depth-2 exchange
    for shape, halo, fuse2 in (((12, 10, 16), 1, 1), ((12, 10, 16), 2, 1), ((20, 16, 64), 2, 2)):
        ctx.set_option("fuse2", fuse2)
        grid, bcs, lat, obcs = hip_cavity_3d(shape, HalfwayBounceBackBC, backend_config={"halo": halo})
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
        f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
        if fuse2 == 2:
            ok &= stepper._native_stepper().step2_eligible(f_0, f_1, bc_mask, missing_mask)
        f_np = orc.perturbed_init(shape, lat, seed=8)
        o_bm, o_mm = orc.build_masks(shape, lat, obcs)
        for overlap in (1, 0):
            ctx.set_option("overlap", overlap)
            f_0.assign(f_np)
            a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.5, 9)
            exp = orc.run(f_np, o_bm, o_mm, obcs, 1.5, lat, 9)
            good = bool(np.array_equal(a.numpy(), exp))
            if not good:
                print(f"mismatch: shape {shape} halo {halo} fuse2 {fuse2} overlap {overlap}", flush=True)
            ok &= good
    stats = ctx.comm_stats()
    if not (stats["halo_waits"] > 0 and stats["halo_wait_ms"] >= 0.0):  # the overlapped runs above timed their halo waits
        print(f"telemetry: {stats}", flush=True)
        ok = False
    print("RCCL_SELF_OK" if ok else "RCCL_SELF_MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
