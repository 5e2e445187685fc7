"""MeshMaskerAABB on the HIP backend vs the oracle's restatement of the reference's
voxelisation (boundary_masker/aabb.py, mesh_boundary_masker.py).  No reference test covers it ("parity unpinned")."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc
from xlb_amd.grid import grid_factory
from xlb_amd.helper import create_nse_fields
from xlb_amd.operator.boundary_condition import FullwayBounceBackBC, HalfwayBounceBackBC, RegularizedBC, ExtrapolationOutflowBC
from xlb_amd.operator.boundary_masker import BC_SOLID, MeshMaskerAABB, MeshVoxelizationMethod
from xlb_amd.operator.stepper import IncompressibleNavierStokesStepper

from _util import init_hip

pytestmark = pytest.mark.gpu


def icosphere(center, radius, subdivisions=2):
    """triangle soup (3 n, 3) of a sphere, built procedurally (no STL reader in this image)"""
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6],
         [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    tris = v[np.array(f)]
    for _ in range(subdivisions):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    tris = tris / np.linalg.norm(tris, axis=2, keepdims=True)
    return (np.asarray(center) + radius * tris).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("lattice", ["D3Q19", "D3Q27"])
def test_aabb_voxelisation_vs_oracle(lattice):
    vs, pp = init_hip(lattice)
    lat = orc.Lattice(lattice)
    shape = (20, 18, 16)
    grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields(shape)
    verts = icosphere((9.3, 8.6, 7.9), 4.2)
    bc = HalfwayBounceBackBC(mesh_vertices=verts, voxelization_method=MeshVoxelizationMethod("AABB"))
    _, bc_mask, missing_mask = MeshMaskerAABB()(bc, f_1, bc_mask, missing_mask)
    assert bc.mesh_vertices is None
    e_bc, e_mm = orc.mesh_mask_aabb(shape, lat, bc.id, verts, np.zeros((1,) + shape, np.uint8), np.zeros((lat.q,) + shape, bool))
    got_bc, got_mm = bc_mask.numpy(), missing_mask.numpy()
    assert np.array_equal(got_bc, e_bc) and np.array_equal(got_mm, e_mm.astype(np.uint8))
    n_solid, n_bnd = int((got_bc == BC_SOLID).sum()), int((got_bc == bc.id).sum())
    assert 150 < n_solid < 600 and n_bnd > n_solid  # a closed shell one voxel thick, fluid boundary voxels on both sides
    # the shell is closed: the sphere's centre cannot be reached from the box corner through non-solid voxels
    from scipy import ndimage

    lab, _ = ndimage.label(got_bc[0] != BC_SOLID)
    assert lab[9, 8, 7] != lab[0, 0, 0]


def test_mesh_outside_the_domain_is_refused():
    vs, pp = init_hip("D3Q19")
    grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields((8, 8, 8))
    bc = HalfwayBounceBackBC(mesh_vertices=icosphere((4, 4, 4), 5.0, 1))
    with pytest.raises(Exception, match="exceed domain dimensions"):
        MeshMaskerAABB()(bc, f_1, bc_mask, missing_mask)
    with pytest.raises(ValueError, match="either indices or mesh_vertices"):
        HalfwayBounceBackBC(mesh_vertices=icosphere((4, 4, 4), 2.0, 1), indices=[[1], [1], [1]])


def test_flow_past_a_mesh_sphere_vs_oracle():
    """The stepper routes a mesh BC through the AABB masker (nse_stepper.py:165-203 in the reference); the run then matches the
    oracle driven with the oracle's masks: profile-free inlet, outflow, fullway walls, halfway wall on the mesh's boundary voxels."""
    shape = (24, 14, 14)
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    box, box_ne = grid.bounding_box_indices(), grid.bounding_box_indices(remove_edges=True)
    walls = [sum((box[f][i] for f in ("bottom", "top", "front", "back")), []) for i in range(3)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    verts = icosphere((8.4, 6.7, 7.2), 2.6, 1)
    b_w = FullwayBounceBackBC(indices=walls)
    b_in = RegularizedBC("velocity", prescribed_value=(0.04, 0.0, 0.0), indices=box_ne["left"])
    b_out = ExtrapolationOutflowBC(indices=box_ne["right"])
    b_s = HalfwayBounceBackBC(mesh_vertices=verts)
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w, b_in, b_out, b_s])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    obcs = [orc.BC(orc.KIND_FULLWAY_BB, b_w.id, walls), orc.BC(orc.KIND_REGULARIZED_VELOCITY, b_in.id, box_ne["left"], prescribed=(0.04, 0.0, 0.0)),
            orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, b_out.id, box_ne["right"])]
    o_bm, o_mm = orc.build_masks(shape, lat, obcs)
    o_bm, o_mm = orc.mesh_mask_aabb(shape, lat, b_s.id, verts, o_bm, o_mm)
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    obcs.append(orc.BC(orc.KIND_HALFWAY_BB, b_s.id, None))
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.4, 20)
    with np.errstate(all="ignore"):
        exp = orc.run(orc.initialize_eq(shape, lat), o_bm, o_mm, obcs, 1.4, lat, 20)
    out = a.numpy()
    fluid = o_bm[0] != orc.BC_SOLID  # solid voxels are never read by anyone: their contents are unspecified
    assert np.array_equal(out[:, fluid], exp[:, fluid])


@pytest.mark.parametrize("lattice", ["D3Q19", "D3Q27"])
def test_ray_voxelisation_vs_oracle(lattice):
    from xlb_amd.operator.boundary_masker import MeshMaskerRay

    vs, pp = init_hip(lattice)
    lat = orc.Lattice(lattice)
    shape = (16, 15, 14)
    grid, f_0, f_1, missing_mask, bc_mask = create_nse_fields(shape)
    verts = icosphere((7.3, 7.6, 6.9), 3.4, 1)
    bc = HalfwayBounceBackBC(mesh_vertices=verts, voxelization_method=MeshVoxelizationMethod("RAY"))
    _, bc_mask, missing_mask = MeshMaskerRay()(bc, f_1, bc_mask, missing_mask)
    e_bc, e_mm = orc.mesh_mask_ray(shape, lat, bc.id, verts, np.zeros((1,) + shape, np.uint8), np.zeros((lat.q,) + shape, bool))
    got_bc, got_mm = bc_mask.numpy(), missing_mask.numpy()
    assert np.array_equal(got_bc, e_bc) and np.array_equal(got_mm, e_mm.astype(np.uint8))
    assert int((got_bc == BC_SOLID).sum()) == 0 and int((got_bc == bc.id).sum()) > 100
    # links are symmetric: if x misses opp(l) because the link x -> x + c_l crosses the surface, x + c_l misses l
    x, y, z = np.where(got_bc[0] == bc.id)
    for l in range(lat.q):
        if l == lat.opp[l]:
            continue
        sel = got_mm[lat.opp[l], x, y, z] == 1
        xn, yn, zn = x[sel] + lat.c[0, l], y[sel] + lat.c[1, l], z[sel] + lat.c[2, l]
        assert np.all(got_mm[l, xn, yn, zn] == 1)


def test_stepper_routes_ray_method():
    shape = (20, 12, 12)
    vs, pp = init_hip("D3Q19")
    lat = orc.Lattice("D3Q19")
    grid = grid_factory(shape)
    verts = icosphere((8.4, 5.7, 6.2), 2.4, 1)
    b_s = HalfwayBounceBackBC(mesh_vertices=verts, voxelization_method=MeshVoxelizationMethod("RAY"))
    stepper = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_s])
    f_0, f_1, bc_mask, missing_mask = stepper.prepare_fields()
    o_bm, o_mm = orc.mesh_mask_ray(shape, lat, b_s.id, verts, np.zeros((1,) + shape, np.uint8), np.zeros((19,) + shape, bool))
    assert np.array_equal(bc_mask.numpy(), o_bm) and np.array_equal(missing_mask.numpy(), o_mm.astype(np.uint8))
    f_np = orc.perturbed_init(shape, lat, seed=61)
    f_0.assign(f_np)
    a, b = stepper.run(f_0, f_1, bc_mask, missing_mask, 1.3, 15)
    exp = orc.run(f_np, o_bm, o_mm, [orc.BC(orc.KIND_HALFWAY_BB, b_s.id, None)], 1.3, lat, 15)
    assert np.array_equal(a.numpy(), exp)
    # the stepper routes every voxelisation method (nse_stepper.py:165-203); WINDING fills the ball
    b_w = HalfwayBounceBackBC(mesh_vertices=verts, voxelization_method=MeshVoxelizationMethod("WINDING"))
    _, _, bm_w, _ = IncompressibleNavierStokesStepper(grid=grid, boundary_conditions=[b_w]).prepare_fields()
    assert (bm_w.numpy() == 255).sum() > 20
    with pytest.raises(AssertionError, match="Unsupported voxelization method"):
        MeshVoxelizationMethod("OCTREE")
