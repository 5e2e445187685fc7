"""Worker of tests/test_gpu_postprocess.py::test_zero_copy_export_to_torch: a fresh process that initialises
PyTorch's HIP runtime FIRST (the wheel bundles its own; libxlbhip.so then binds to the one already loaded — the other
order leaves torch with "No HIP GPUs are available")."""

import os
import sys

import numpy as np
import torch

torch.cuda.init()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import init_hip  # noqa: E402
from xlb_amd.grid import grid_factory  # noqa: E402
from xlb_amd.precision_policy import Precision  # noqa: E402


def main():
    init_hip("D3Q19")
    for cfg, shape in ((None, (6, 8, 16)), ({"halo": 2}, (6, 8, 16))):
        grid = grid_factory(shape, backend_config=cfg)
        f = grid.create_field(19)
        ref = np.random.default_rng(1).random((19,) + shape).astype(np.float32)
        f.assign(ref)
        t = torch.from_dlpack(f)
        assert t.is_cuda and tuple(t.shape) == (19,) + shape and t.dtype == torch.float32
        assert np.array_equal(t.cpu().numpy(), ref)
        t2 = torch.as_tensor(f, device="cuda")  # CUDA array interface
        assert t2.data_ptr() == t.data_ptr() and np.array_equal(t2.cpu().numpy(), ref)
        # aliasing, both ways
        t[3, 1, 2, 5] = 42.0
        torch.cuda.synchronize()
        assert f.numpy()[3, 1, 2, 5] == 42.0
        f.fill(0.5)
        f.ctx.sync()
        assert float(t.sum().item()) == 0.5 * ref.size
        # a reduction the backend has no operator for, straight on the field's memory: total mass
        f.assign(ref)
        assert abs(float(t.double().sum().item()) - float(ref.astype(np.float64).sum())) < 1e-6
    init_hip("D2Q9")
    g2 = grid_factory((8, 16))
    m = g2.create_field(1, dtype=Precision.UINT8, fill_value=3)
    tm = torch.from_dlpack(m)
    assert tuple(tm.shape) == (1, 8, 16) and tm.dtype == torch.uint8 and int(tm.sum().item()) == 3 * 128
    edited_mask_is_seen()
    print("DLPACK_OK")  # (t, t2, tm die at interpreter shutdown: the deleter thunk is built to survive that)


def edited_mask_is_seen():
    """the stepper caches the two-step kernel's meta words / clean flags on the masks' contents
    version; a mask edited through its zero-copy alias goes through no C-ABI writer.  An exported mask therefore counts
    as modified on every use (xlbhip_field_touch): the second run must see the edit."""
    from oracle import xlb_numpy as orc
    from xlb_amd.default_config import get_context
    from xlb_amd.operator.boundary_condition import FullwayBounceBackBC
    from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

    from _util import hip_cavity_3d

    shape = (12, 16, 64)
    grid, bcs, lat, obcs = hip_cavity_3d(shape, FullwayBounceBackBC)
    get_context().set_option("fuse2", 2)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    f_0, f_1, bm, mm = stepper.prepare_fields()
    f_np = orc.perturbed_init(shape, lat, seed=3)
    f_0.assign(f_np)
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    t_bm = torch.as_tensor(bm, device="cuda")  # writable alias of bc_mask
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, 1.3, 4)  # pairs: meta words cached for these masks
    e = orc.run(f_np, o_bm, o_mm, obcs, 1.3, lat, 4)
    assert np.array_equal(f_0.numpy(), e)
    # open a window in the y = 0 wall: those cells become plain fluid (periodic wrap), in the oracle's mask too
    t_bm[0, 3:9, 0, 20:40] = 0
    torch.cuda.synchronize()
    o_bm2 = o_bm.copy()
    o_bm2[0, 3:9, 0, 20:40] = 0
    assert np.array_equal(bm.numpy(), o_bm2)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, 1.3, 4)
    assert np.array_equal(f_0.numpy(), orc.run(e, o_bm2, o_mm, obcs, 1.3, lat, 4)), "edit of an exported bc_mask was not seen"
    # ... and a POPULATION field edited through its alias: the two-step kernel keeps a strip buffer per field (the halo columns of
    # its tiles), a cache of the field's contents; an exported field counts as modified on every use, so the strips are not trusted.
    # The edit sits on a tile boundary of the (8 x 64) tiling with half-tile shift (z = 31 | 32), where stale strips would be pulled.
    state = f_0.numpy()
    t_f = torch.as_tensor(f_0, device="cuda")
    t_f[:, 2:9, 3:11, 30:34] *= 1.001
    torch.cuda.synchronize()
    state[:, 2:9, 3:11, 30:34] *= np.float32(1.001)
    assert np.array_equal(f_0.numpy(), state)
    f_0, f_1 = stepper.run(f_0, f_1, bm, mm, 1.3, 4)
    assert np.array_equal(f_0.numpy(), orc.run(state, o_bm2, o_mm, obcs, 1.3, lat, 4)), "edit of an exported population field was not seen (stale strips)"
    get_context().set_option("fuse2", 1)


if __name__ == "__main__":
    main()
