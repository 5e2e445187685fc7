"""Vorticity / QCriterion on the HIP backend vs the oracle's restatement of the
reference's kernels (postprocess/vorticity.py:30-84, q_criterion.py:36-131).  No reference test covers them
("parity unpinned by the reference"); bit-exact against the oracle, plus an analytic field."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.operator.postprocess import QCriterion, Vorticity
from xlb_amd.precision_policy import Precision

from _util import hip_cavity_3d, init_hip
from xlb_amd.operator.boundary_condition import HalfwayBounceBackBC
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy,prec", [("FP32FP32", Precision.FP32), ("FP64FP64", Precision.FP64)])
def test_vorticity_and_q_vs_oracle(policy, prec):
    shape = (12, 10, 14)
    grid, bcs, lat, obcs = hip_cavity_3d(shape, HalfwayBounceBackBC, policy=policy)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=bcs)
    _, _, bc_mask, _ = stepper.prepare_fields()
    T = prec.np_dtype
    rng = np.random.default_rng(5)
    u_np = (0.1 * rng.standard_normal((3,) + shape)).astype(T)
    u = grid.create_field(3, dtype=prec).assign(u_np)
    vort = grid.create_field(3, dtype=prec, fill_value=7.0)  # sentinel: untouched cells must keep it
    mag = grid.create_field(1, dtype=prec, fill_value=7.0)
    norm_mu = grid.create_field(1, dtype=prec, fill_value=7.0)
    q = grid.create_field(1, dtype=prec, fill_value=7.0)
    bm = bc_mask.numpy()
    v_out, m_out = Vorticity()(u, bc_mask, vort, mag)
    n_out, q_out = QCriterion()(u, bc_mask, norm_mu, q)
    sent = np.full((3,) + shape, 7.0, T)
    e_v, e_m = orc.vorticity(u_np, bm, sent, sent[:1])
    e_n, e_q = orc.q_criterion(u_np, bm, sent[:1], sent[:1])
    assert np.array_equal(v_out.numpy(), e_v) and np.array_equal(m_out.numpy(), e_m)
    assert np.array_equal(n_out.numpy(), e_n) and np.array_equal(q_out.numpy(), e_q)
    # the cavity's boundary layer and the cells next to it are skipped, the core is written
    assert np.all(e_m[0, :2] == 7.0) and np.all(e_m[0, 2:-2, 2:-2, 2:-2] != 7.0)


def test_rigid_rotation_and_shear():
    """u = Omega x r has vorticity 2 Omega and Q = |Omega|^2 everywhere; u = (g y, 0, 0) has Q = -g^2 / 4... exactly
    representable inputs, so central differences are exact up to rounding."""
    vs, pp = init_hip("D3Q19")
    from xlb_amd.grid import grid_factory

    shape = (10, 12, 8)
    grid = grid_factory(shape)
    bc_mask = grid.create_field(1, dtype=Precision.UINT8)
    x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float32) for n in shape], indexing="ij")
    om = np.array([0.25, -0.5, 0.125], np.float32)
    u_np = np.stack([om[1] * z - om[2] * y, om[2] * x - om[0] * z, om[0] * y - om[1] * x]).astype(np.float32)
    u = grid.create_field(3, dtype=Precision.FP32).assign(u_np)
    vort, mag = Vorticity()(u, bc_mask, grid.create_field(3, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
    core = (slice(None), slice(1, -1), slice(1, -1), slice(1, -1))
    assert np.allclose(vort.numpy()[core], (2 * om)[:, None, None, None], atol=1e-6)
    _, q = QCriterion()(u, bc_mask, grid.create_field(1, dtype=Precision.FP32), grid.create_field(1, dtype=Precision.FP32))
    assert np.allclose(q.numpy()[core], float(om @ om), atol=1e-6)
    assert np.all(q.numpy()[0, 0] == 0) and np.all(mag.numpy()[0, :, :, -1] == 0)  # the outer layer is never written


def test_zero_copy_export_descriptors():
    """Field.__dlpack__ / __cuda_array_interface__ describe the interior of the field as a strided device array (padded
    plane stride and ghost planes expressed by strides): the zero-copy replacement of the reference's ToJAX /
    warp_array_to_jax copies.  Checked structurally (capsule contents read back through ctypes, data fetched with the
    described strides); test_zero_copy_export_to_torch hands them to a real consumer."""
    import ctypes as C

    from xlb_amd import _lib
    from xlb_amd.grid import grid_factory

    vs, pp = init_hip("D3Q19")
    for cfg, shape in ((None, (6, 8, 16)), ({"halo": 2}, (6, 8, 16))):
        grid = grid_factory(shape, backend_config=cfg)
        f = grid.create_field(19)
        ref = np.random.default_rng(1).random((19,) + shape).astype(np.float32)
        f.assign(ref)
        info = f.info()
        cai = f.__cuda_array_interface__
        halo = 0 if cfg is None else 2
        assert cai["shape"] == (19,) + shape and cai["typestr"] == "<f4" and cai["version"] == 3
        assert cai["data"] == (info["device_ptr"] + halo * 8 * 16 * 4, False)
        assert cai["strides"] == (info["plane_stride"] * 4, 8 * 16 * 4, 16 * 4, 4)
        assert f.__dlpack_device__() == (10, f.ctx.device)  # kDLROCM
        cap = f.__dlpack__()
        C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
        C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
        m = C.cast(C.pythonapi.PyCapsule_GetPointer(cap, b"dltensor"), C.POINTER(_lib._DLManagedTensor)).contents
        t = m.dl_tensor
        assert t.ndim == 4 and [t.shape[i] for i in range(4)] == [19, *shape]
        assert [t.strides[i] for i in range(4)] == [info["plane_stride"], 8 * 16, 16, 1] and t.byte_offset == 0
        assert (t.dtype.code, t.dtype.bits, t.dtype.lanes) == (2, 32, 1) and (t.device.device_type, t.device.device_id) == (10, f.ctx.device)
        assert t.data == cai["data"][0]
        n_alive = len(_lib._dl_alive)
        del m, t, cap  # an unconsumed capsule releases its tensor when it dies
        import gc

        gc.collect()
        assert len(_lib._dl_alive) == n_alive - 1
        # the described view really is the data: population 7, plane 3 through the storage-plane accessor
        assert np.array_equal(f.get_plane(7, 3 + halo), ref[7, 3])
    with pytest.raises(Exception, match="missing_mask"):
        grid.create_missing_mask(19).__dlpack__()


def test_zero_copy_export_to_torch():
    """torch.from_dlpack(field) / torch.as_tensor(field, device="cuda") alias the field's HBM, both directions.  Fresh
    process with torch initialised first (see the worker's docstring for why the order matters)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "_gpu_dlpack_worker.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DLPACK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]



@pytest.mark.parametrize("policy", ["FP32FP32", "FP64FP64"])
def test_momentum_transfer_on_a_sphere(policy):
    """MomentumTransfer(no_slip_bc)(f_0, f_1, bc_mask, missing_mask) on a halfway sphere in a driven channel, against the
    oracle's restatement of force/momentum_transfer.py:167-205.  The per-cell terms are the same arithmetic; the grid sum
    is accumulated in double on the device (the reference's order is XLA's), hence a tolerance instead of bit-equality."""
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, RegularizedBC
    from xlb_amd.operator.force import MomentumTransfer

    shape = (24, 12, 12)
    vs, pp = init_hip("D3Q19", policy)
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    walls = [sum((box[f][i] for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    sphere = [s.tolist() for s in np.where((x - 8) ** 2 + (y - 6) ** 2 + (z - 6) ** 2 < 2.6**2)]
    b_w = FullwayBounceBackBC(indices=walls)
    b_in = RegularizedBC("velocity", prescribed_value=(0.04, 0.0, 0.0), indices=box_ne["left"])
    b_out = RegularizedBC("pressure", prescribed_value=1.0, indices=box_ne["right"])
    b_s = HalfwayBounceBackBC(indices=sphere)
    obcs = [orc.BC(orc.KIND_FULLWAY_BB, b_w.id, walls), orc.BC(orc.KIND_REGULARIZED_VELOCITY, b_in.id, box_ne["left"], prescribed=(0.04, 0.0, 0.0)),
            orc.BC(orc.KIND_REGULARIZED_PRESSURE, b_out.id, box_ne["right"], prescribed=1.0), orc.BC(orc.KIND_HALFWAY_BB, b_s.id, sphere)]
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_in, b_out, b_s])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.4, 60)
    force = MomentumTransfer(b_s)(f_0, f_1, bc_mask, missing_mask)
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    exp = orc.momentum_transfer(f_0.numpy(), obcs[3], o_bm, o_mm, lat, policy)
    assert force.shape == (3,) and force.dtype == orc.compute_dtype(policy)
    tol = 1e-5 if policy == "FP32FP32" else 1e-12
    assert np.allclose(force, exp, rtol=tol, atol=tol * np.abs(exp).max()), (force, exp)
    assert force[0] > 0 and abs(force[1]) < 0.05 * force[0] and abs(force[2]) < 0.05 * force[0]  # drag along the flow, no net lift
    # a fullway no-slip BC goes through the same operator (f_post_stream[l] = f_0[opp l])
    fw = MomentumTransfer(b_w)(f_0, f_1, bc_mask, missing_mask)
    exp_fw = orc.momentum_transfer(f_0.numpy(), obcs[0], o_bm, o_mm, lat, policy)
    assert np.allclose(fw, exp_fw, rtol=10 * tol, atol=10 * tol * np.abs(exp_fw).max()), (fw, exp_fw)


@pytest.mark.parametrize("prec", [Precision.FP32, Precision.FP64])
def test_grid_to_point_vs_oracle(prec):
    from xlb_amd.grid import grid_factory
    from xlb_amd.operator.postprocess import GridToPoint

    vs, pp = init_hip("D3Q19")
    shape = (9, 7, 11)
    grid = grid_factory(shape)
    T = prec.np_dtype
    rng = np.random.default_rng(4)
    g_np = rng.standard_normal((1,) + shape).astype(T)
    fld = grid.create_field(1, dtype=prec).assign(g_np)
    pts = (rng.random((200, 3)) * (np.array(shape) - 1.001)).astype(np.float32)
    pts[:3] = [[0, 0, 0], [2.0, 3.0, 4.0], [7.5, 5.25, 9.75]]  # grid nodes reproduce the node value exactly
    out = GridToPoint()(fld, pts, np.empty(200, T))
    assert np.array_equal(out, orc.grid_to_point(g_np, pts))
    assert out[0] == g_np[0, 0, 0, 0] and out[1] == g_np[0, 2, 3, 4]
    # a linear field is reproduced by trilinear interpolation
    x, y, z = np.meshgrid(*[np.arange(n, dtype=T) for n in shape], indexing="ij")
    lin = (0.5 * x - 0.25 * y + 2.0 * z + 1.0)[None]
    fld.assign(lin)
    out = GridToPoint()(fld, pts, np.empty(200, T))
    assert np.allclose(out, 0.5 * pts[:, 0] - 0.25 * pts[:, 1] + 2.0 * pts[:, 2] + 1.0, atol=1e-4)
    with pytest.raises(Exception, match="outside the field"):
        GridToPoint()(fld, np.array([[8.5, 1.0, 1.0]], np.float32), np.empty(1, T))
