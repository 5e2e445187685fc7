"""SURVEY 8(f) rank 4, the rest: MeshMaskerWinding, MeshMaskerAABBClose, wall distances and HybridBC on the HIP backend vs
the oracle's restatement (oracle/mesh_bc.py).  Reference: boundary_masker/{ray,winding,aabb_close}.py, bc_hybrid.py — kernel
backends only, no reference test ("parity unpinned by the reference")."""

import numpy as np
import pytest

from oracle import mesh_bc as mb
from oracle import xlb_numpy as orc
from xlb_amd.grid import grid_factory
from xlb_amd.helper import create_nse_fields
from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, HalfwayBounceBackBC, HybridBC
from xlb_amd.operator.boundary_masker import BC_SOLID, MeshMaskerAABB, MeshVoxelizationMethod, mesh_masker_for
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import icosphere, init_hip

pytestmark = pytest.mark.gpu
SHAPE = (18, 16, 14)
CENTER, RADIUS = (8.3, 7.6, 6.9), 3.7
KINDS = {"bounceback_regularized": mb.KIND_HYBRID_BB_REGULARIZED, "bounceback_grads": mb.KIND_HYBRID_BB_GRADS,
         "nonequilibrium_regularized": mb.KIND_HYBRID_NEQ_REGULARIZED}


def oracle_mask(method, lat, bc_id, verts, with_dist, close_voxels=2):
    z1, zq = np.zeros((1,) + SHAPE, np.uint8), np.zeros((lat.q,) + SHAPE, bool)
    d0 = np.zeros((lat.q,) + SHAPE, np.float32) if with_dist else None
    if method == "RAY":
        return mb.mesh_mask_ray(SHAPE, lat, bc_id, verts, z1, zq, d0)
    if method == "WINDING":
        return mb.mesh_mask_winding(SHAPE, lat, bc_id, verts, z1, zq, d0)
    return mb.mesh_mask_aabb_close(SHAPE, lat, bc_id, verts, close_voxels, z1, zq, d0)


def method_of(name):
    return MeshVoxelizationMethod(name, close_voxels=2) if name == "AABB_CLOSE" else MeshVoxelizationMethod(name)


@pytest.mark.parametrize("lattice", ["D3Q19", "D3Q27"])
@pytest.mark.parametrize("method", ["RAY", "WINDING", "AABB_CLOSE"])
@pytest.mark.parametrize("with_dist", [False, True])
def test_voxelisation_and_distances_vs_oracle(lattice, method, with_dist):
    vs, pp = init_hip(lattice)
    lat = orc.Lattice(lattice)
    grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields(SHAPE)
    verts = icosphere(CENTER, RADIUS, 1)
    bc = HybridBC("bounceback_regularized", mesh_vertices=verts, voxelization_method=method_of(method), use_mesh_distance=with_dist)
    masker = mesh_masker_for(bc.voxelization_method)
    ret, bc_mask, missing_mask = masker(bc, f_1, bc_mask, missing_mask)
    assert ret is f_1 and bc.mesh_vertices is None  # the distances argument (the reference passes f_1) is handed back untouched
    e_bc, e_mm, e_d = oracle_mask(method, lat, bc.id, verts, with_dist)
    assert np.array_equal(bc_mask.numpy(), e_bc) and np.array_equal(missing_mask.numpy(), e_mm.astype(np.uint8))
    assert not f_1.numpy().any()
    if with_dist:
        cells, w = bc._distance_table
        exp_cells = np.flatnonzero(e_bc.reshape(-1) == bc.id)
        assert np.array_equal(cells, exp_cells) and w.shape == (len(cells), lat.q) and w.dtype == np.float32
        assert np.array_equal(w, e_d.reshape(lat.q, -1)[:, exp_cells].T)  # bit for bit (no square roots on either side)
        assert w.max() <= 1.0 and (w != 0).sum() > 100
    else:
        assert bc._distance_table is None
    if method != "RAY":
        solid = bc_mask.numpy()[0] == BC_SOLID
        assert solid[8, 7, 6] and 100 < solid.sum() < 500  # the ball is filled


def test_masker_argument_checks():
    vs, pp = init_hip("D3Q19")
    grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields((12, 12, 12))
    verts = icosphere((6, 6, 6), 3.0, 1)
    with pytest.raises(Exception, match="no wall distances"):
        bc = HybridBC("bounceback_grads", mesh_vertices=verts, use_mesh_distance=True)
        MeshMaskerAABB()(bc, f_1, bc_mask, missing_mask)
    with pytest.raises(AssertionError, match="close voxels"):
        mesh_masker_for(MeshVoxelizationMethod("AABB_CLOSE"))
    with pytest.raises(Exception, match="exceed domain dimensions"):
        bc = HybridBC("bounceback_grads", mesh_vertices=icosphere((6, 6, 6), 7.0, 1), voxelization_method=MeshVoxelizationMethod("WINDING"))
        mesh_masker_for(bc.voxelization_method)(bc, f_1, bc_mask, missing_mask)
    with pytest.raises(AssertionError, match="mesh vertices"):
        HybridBC("bounceback_grads", indices=[[1], [1], [1]], use_mesh_distance=True)
    with pytest.raises(AssertionError, match="not supported"):
        HybridBC("bounce", indices=[[1], [1], [1]])
    init_hip("D2Q9")
    with pytest.raises(NotImplementedError, match="2D"):
        HybridBC("bounceback_grads", indices=[[1], [1]])


def body_indices(shape):
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    return np.array(np.where((x - shape[0] / 2 + 0.3) ** 2 + (y - shape[1] / 2) ** 2 + (z - shape[2] / 2 + 0.4) ** 2 < 2.4**2))


@pytest.mark.parametrize("bc_method", list(KINDS))
@pytest.mark.parametrize("lattice,policy,u_wall", [("D3Q19", "FP32FP32", None), ("D3Q19", "FP32FP32", (0.02, -0.01, 0.005)),
                                                  ("D3Q27", "FP32FP32", None), ("D3Q19", "FP64FP64", (0.0, 0.03, 0.0))])
def test_hybrid_bc_on_indices_vs_oracle(bc_method, lattice, policy, u_wall):
    """A body given by interior indices (padded like a halfway wall) in a periodic box with a mean flow, 8 steps: fused step
    kernel (extended-BC variant) vs the oracle, bit for bit."""
    vs, pp = init_hip(lattice, policy)
    lat = orc.Lattice(lattice)
    shape = (12, 10, 10)
    grid = grid_factory(shape)
    body = body_indices(shape)
    bc = HybridBC(bc_method, prescribed_value=u_wall, indices=body.tolist())
    obc = mb.HybridBC(KINDS[bc_method], bc.id, body, u_wall=u_wall)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.build_masks(shape, lat, [obc])
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    f_np = orc.perturbed_init(shape, lat, policy, seed=41, amp_rho=0.01, amp_u=0.03)
    f_0.assign(f_np)
    steps, omega = 8, 1.5
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    exp = mb.run(f_np, o_bm, o_mm, [obc], omega, lat, steps, policy)
    out = f_0.numpy()
    assert np.isfinite(out).all()
    assert np.abs(out.astype(np.float64) - exp.astype(np.float64)).max() <= 1e-6
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("bc_method", list(KINDS))
@pytest.mark.parametrize("method", ["RAY", "WINDING", "AABB_CLOSE"])
def test_hybrid_bc_on_a_mesh_with_distances_vs_oracle(bc_method, method):
    """Flow past the mesh sphere with the curved-wall interpolation (use_mesh_distance), fullway channel walls around it:
    masks, distance table and 10 steps vs the oracle."""
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(SHAPE)
    verts = icosphere(CENTER, RADIUS, 1)
    box = grid.bounding_box_indices()
    walls = [box["bottom"][i] + box["top"][i] for i in range(3)]
    b_w = FullwayBounceBackBC(indices=walls)
    b_s = HybridBC(bc_method, mesh_vertices=verts, voxelization_method=method_of(method), use_mesh_distance=True)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_s])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_w = orc.BC(orc.KIND_FULLWAY_BB, b_w.id, walls)
    o_bm, o_mm = orc.build_masks(SHAPE, lat, [o_w])
    z = np.zeros((lat.q,) + SHAPE, np.float32)
    if method == "RAY":
        o_bm, o_mm, o_d = mb.mesh_mask_ray(SHAPE, lat, b_s.id, verts, o_bm, o_mm, z)
    elif method == "WINDING":
        o_bm, o_mm, o_d = mb.mesh_mask_winding(SHAPE, lat, b_s.id, verts, o_bm, o_mm, z)
    else:
        o_bm, o_mm, o_d = mb.mesh_mask_aabb_close(SHAPE, lat, b_s.id, verts, 2, o_bm, o_mm, z)
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    o_s = mb.HybridBC(KINDS[bc_method], b_s.id, None, distances=o_d)
    f_np = orc.perturbed_init(SHAPE, lat, seed=43, amp_rho=0.01, amp_u=0.02)
    f_0.assign(f_np)
    steps, omega = 10, 1.3
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    with np.errstate(all="ignore"):
        exp = mb.run(f_np, o_bm, o_mm, [o_w, o_s], omega, lat, steps)
    out = f_0.numpy()
    fluid = np.broadcast_to(o_bm != BC_SOLID, out.shape)  # (solid voxels are never read by the fluid; compare them too if finite)
    assert np.isfinite(out[fluid]).all()
    assert np.array_equal(out[fluid], exp[fluid])


@pytest.mark.parametrize("bc_method", list(KINDS))
@pytest.mark.parametrize("with_dist", [False, True])
@pytest.mark.parametrize("lattice,policy", [("D3Q19", "FP32FP32"), ("D3Q27", "FP64FP32")])
def test_hybrid_bc_wall_velocity_profile_vs_oracle(bc_method, with_dist, lattice, policy):
    """A ROTATING mesh sphere (the reference's examples/cfd/rotating_sphere_3d.py:114-146): HybridBC(profile=...) with the wall velocity
    omega x (r - centre) per boundary cell — the reference evaluates profile(index) in the kernel (bc_hybrid.py:265), here the stepper
    evaluates the callable at the BC's cells and the kernel reads a sparse table.  Masks and 8 steps vs the oracle, bit for bit."""
    vs, pp = init_hip(lattice, policy)
    lat = orc.Lattice(lattice)
    grid = grid_factory(SHAPE)
    verts = icosphere(CENTER, RADIUS, 1)
    rot = np.array([0.0, 0.004, -0.002])
    seen = {}

    def profile(cells):
        seen["n"] = cells.shape[1]
        r = cells.astype(np.float64) - np.asarray(CENTER).reshape(3, 1)
        return np.cross(rot.reshape(1, 3), r.T).T

    b_s = HybridBC(bc_method, profile=profile, mesh_vertices=verts, voxelization_method=method_of("RAY"), use_mesh_distance=with_dist)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_s], collision_type="KBC" if lattice == "D3Q27" else "BGK")
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    z1, zq = np.zeros((1,) + SHAPE, np.uint8), np.zeros((lat.q,) + SHAPE, bool)
    res = mb.mesh_mask_ray(SHAPE, lat, b_s.id, verts, z1, zq, np.zeros((lat.q,) + SHAPE, np.float32) if with_dist else None)
    o_bm, o_mm = res[0], res[1]
    o_d = res[2] if with_dist else None
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    assert seen["n"] == int((o_bm[0] == b_s.id).sum()) > 0
    # the same field of wall velocities for the oracle (values only matter at the BC's cells)
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in SHAPE], indexing="ij")).reshape(3, -1)
    uw = profile(idx).reshape((3,) + SHAPE)
    o_s = mb.HybridBC(KINDS[bc_method], b_s.id, None, u_wall=uw, distances=o_d)
    f_np = orc.perturbed_init(SHAPE, lat, policy, seed=47, amp_rho=0.01, amp_u=0.02)
    f_0.assign(f_np)
    steps, omega = 8, 1.4
    ctx = __import__("xlb_amd.default_config", fromlist=["get_context"]).get_context()
    try:
        ctx.set_option("exact_math", 1)
        f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
        out = f_0.numpy()
    finally:
        ctx.set_option("exact_math", 0)
    with np.errstate(all="ignore"):
        exp = mb.run(f_np, o_bm, o_mm, [o_s], omega, lat, steps, policy, "KBC" if lattice == "D3Q27" else "BGK")
    fluid = np.broadcast_to(o_bm != BC_SOLID, out.shape)
    assert np.isfinite(out[fluid]).all()
    assert np.array_equal(out[fluid], exp[fluid])
    # the force on the sphere (MomentumTransfer through the stepper's distance / velocity tables) vs the oracle; the per-cell terms are
    # the same arithmetic, the grid sum's order is not (atomics in the reference as well)
    from xlb_amd.operator.force import MomentumTransfer

    force = MomentumTransfer(b_s)(f_0, f_1, bc_mask, missing_mask)
    with np.errstate(all="ignore"):
        fexp = mb.momentum_transfer(np.where(fluid, out, 0).astype(out.dtype), o_s, o_bm, o_mm, lat, policy)
    tol = 2e-5 if policy == "FP32FP32" else 1e-11
    assert force.shape == (3,) and np.allclose(force, fexp, rtol=tol, atol=tol * np.abs(fexp).max()), (force, fexp)
    assert np.abs(fexp).max() > 0
    with pytest.raises(Exception, match="runs inside the stepper"):
        b_s(f_0, f_1, bc_mask, missing_mask)  # the stand-alone operator call has no table to read


@pytest.mark.parametrize("case", ["mesh_3d", "indices_2d"])
def test_halfway_bc_wall_velocity_profile_vs_oracle(case):
    """HalfwayBounceBackBC(profile=...) (the alternative the reference's rotating-sphere driver names, rotating_sphere_3d.py:137): the wall
    velocity omega x (r - centre) per boundary cell, on a RAY-voxelised mesh sphere in 3-D and on an index-built cylinder in 2-D."""
    if case == "mesh_3d":
        vs, pp = init_hip("D3Q19")
        lat, shape = orc.Lattice("D3Q19"), SHAPE
        grid = grid_factory(shape)
        rot, ctr = np.array([0.003, 0.0, -0.004]), np.asarray(CENTER)

        def profile(cells):
            return np.cross(rot.reshape(1, 3), (cells.astype(np.float64) - ctr.reshape(3, 1)).T).T

        bc = HalfwayBounceBackBC(profile=profile, mesh_vertices=icosphere(CENTER, RADIUS, 1), voxelization_method=method_of("RAY"))
        z1, zq = np.zeros((1,) + shape, np.uint8), np.zeros((lat.q,) + shape, bool)
        o_bm, o_mm = mb.mesh_mask_ray(shape, lat, bc.id, icosphere(CENTER, RADIUS, 1), z1, zq, None)[:2]
    else:
        vs, pp = init_hip("D2Q9")
        lat, shape = orc.Lattice("D2Q9"), (24, 20)
        grid = grid_factory(shape)
        ctr, w = np.array([11.3, 9.6]), 0.005
        yy, zz = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
        inside = (yy - ctr[0]) ** 2 + (zz - ctr[1]) ** 2 <= 4.2**2
        idx = [yy[inside].tolist(), zz[inside].tolist()]

        def profile(cells):  # rotation about the axis normal to the plane: u = w (-(z - cz), (y - cy))
            r = cells.astype(np.float64) - ctr.reshape(2, 1)
            return np.stack([-w * r[1], w * r[0]])

        bc = HalfwayBounceBackBC(profile=profile, indices=idx)
        o_bm, o_mm = orc.build_masks(shape, lat, [orc.BC(orc.KIND_HALFWAY_BB, bc.id, idx)])
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    allc = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")).reshape(lat.d, -1)
    uw = profile(allc).reshape((lat.d,) + shape)
    o_b = mb.HalfwayProfileBC(bc.id, None, uw)
    f_np = orc.perturbed_init(shape, lat, seed=53, amp_rho=0.01, amp_u=0.02)
    f_0.assign(f_np)
    steps, omega = 9, 1.5
    f_0, f_1 = stepper.run(f_0, f_1, bc_mask, missing_mask, omega, steps)
    with np.errstate(all="ignore"):
        exp = mb.run(f_np, o_bm, o_mm, [o_b], omega, lat, steps)
    out = f_0.numpy()
    fluid = np.broadcast_to(o_bm != BC_SOLID, out.shape)
    assert np.isfinite(out[fluid]).all()
    assert np.array_equal(out[fluid], exp[fluid])
    # the wall really moves: the same run with a no-slip wall differs
    with np.errstate(all="ignore"):
        still = mb.run(f_np, o_bm, o_mm, [mb.HalfwayProfileBC(bc.id, None, np.zeros_like(uw))], omega, lat, steps)
    assert not np.array_equal(still[fluid], exp[fluid])
    from xlb_amd.operator.force import MomentumTransfer

    force = MomentumTransfer(bc)(f_0, f_1, bc_mask, missing_mask)
    with np.errstate(all="ignore"):
        fexp = mb.momentum_transfer(np.where(fluid, out, 0).astype(out.dtype), o_b, o_bm, o_mm, lat)
    assert force.shape == (lat.d,) and np.allclose(force, fexp, rtol=2e-5, atol=2e-5 * np.abs(fexp).max()), (force, fexp)


def test_hybrid_bc_standalone_operator_vs_oracle():
    """bc(f_pre, f_post, bc_mask, missing_mask) -> f_post (boundary_condition.py:146-180), without mesh distances."""
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    shape = (10, 9, 8)
    grid = grid_factory(shape)
    body = body_indices(shape)
    for name, kind in KINDS.items():
        bc = HybridBC(name, prescribed_value=(0.01, 0.0, -0.02), indices=body.tolist())
        obc = mb.HybridBC(kind, bc.id, body, u_wall=(0.01, 0.0, -0.02))
        stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[bc])
        f_pre, f_post, bc_mask, missing_mask = stepper.prepare_fields()
        pre = orc.perturbed_init(shape, lat, seed=5, amp_u=0.03)
        post = orc.stream(pre, lat)
        f_pre.assign(pre)
        f_post.assign(post)
        out = bc(f_pre, f_post, bc_mask, missing_mask)
        o_bm, o_mm = orc.build_masks(shape, lat, [obc])
        exp = mb.apply_hybrid(obc, pre, post, o_bm, o_mm, lat, "FP32FP32")
        assert np.array_equal(out.numpy(), exp)
