"""oracle/mesh_bc.py (WINDING / AABB_CLOSE voxelisation, wall distances, HybridBC): nothing in the reference pins these
(kernel backends only, no tests — "parity unpinned by the reference"), so the restatement is held by properties that follow
from the reference's formulas."""

import numpy as np
import pytest

from oracle import mesh_bc as mb
from oracle import xlb_numpy as orc

from _util import icosphere

SHAPE = (16, 14, 14)
CENTER, RADIUS = (7.3, 6.6, 6.9), 3.6


@pytest.fixture(scope="module")
def masks():
    lat = orc.Lattice("D3Q19")
    verts = icosphere(CENTER, RADIUS, 1)
    z1, zq, d0 = np.zeros((1,) + SHAPE, np.uint8), np.zeros((lat.q,) + SHAPE, bool), np.zeros((lat.q,) + SHAPE, np.float32)
    return lat, verts, {
        "ray": mb.mesh_mask_ray(SHAPE, lat, 3, verts, z1, zq, d0),
        "winding": mb.mesh_mask_winding(SHAPE, lat, 3, verts, z1, zq, d0),
        "close": mb.mesh_mask_aabb_close(SHAPE, lat, 3, verts, 2, z1, zq, d0),
        "aabb": orc.mesh_mask_aabb(SHAPE, lat, 3, verts, z1, zq),
    }


def test_winding_number_of_a_closed_surface():
    verts = icosphere(CENTER, RADIUS, 1).reshape(-1, 3, 3)
    assert abs(mb.winding_number(np.array(CENTER), verts) - 1.0) < 1e-12
    assert abs(mb.winding_number(np.array([1.0, 1.0, 1.0]), verts)) < 1e-12
    assert abs(mb.winding_number(np.array(CENTER), verts[:, ::-1]) + 1.0) < 1e-12  # inward orientation: -1 (Warp: outside)


def test_winding_solid_is_the_ball(masks):
    lat, verts, m = masks
    bc = m["winding"][0]
    x, y, z = np.meshgrid(*[np.arange(n) + 0.5 for n in SHAPE], indexing="ij")
    r = np.sqrt((x - CENTER[0]) ** 2 + (y - CENTER[1]) ** 2 + (z - CENTER[2]) ** 2)
    solid = bc[0] == mb.BC_SOLID
    assert solid[r < RADIUS * 0.93].all() and not solid[r > RADIUS].any()  # (the icosphere is inscribed: facets cut a little off)
    # every boundary voxel is fluid, touches a solid voxel along one of its missing directions, and no solid voxel is tagged
    bnd = bc[0] == 3
    assert bnd.any() and not (bnd & solid).any()
    mm = m["winding"][1]
    for l in range(lat.q):
        c = lat.c[:, l]
        src = np.roll(solid, shift=tuple(int(v) for v in c), axis=(0, 1, 2))  # solid at x - c_l
        assert not (mm[l] & ~src).any()  # missing[l] only where the pull x - c_l comes out of a solid voxel


def test_ray_and_winding_agree_on_the_distances(masks):
    """A link between an outside voxel A and an inside voxel B is cut once: the ray from A (RAY) and the ray from B (WINDING)
    see the same point, and both methods store the fraction measured from A in the same slot."""
    lat, verts, m = masks
    d_ray, d_win = m["ray"][2], m["winding"][2]
    both = (d_ray > 0) & (d_win > 0)
    assert both.sum() > 200
    assert np.abs(d_ray[both] - d_win[both]).max() < 1e-5
    assert d_ray[d_ray > 0].min() > 0 and d_ray.max() <= 1.0


def test_ray_masks_unchanged_by_the_distance_output(masks):
    lat, verts, m = masks
    bc0, mm0 = orc.mesh_mask_ray(SHAPE, lat, 3, verts, np.zeros((1,) + SHAPE, np.uint8), np.zeros((lat.q,) + SHAPE, bool))
    assert np.array_equal(m["ray"][0], bc0) and np.array_equal(m["ray"][1], mm0)
    # a weight exactly where the opposite missing bit is set by a cut link
    for l in range(lat.q):
        if l != lat.opp[l]:
            assert np.array_equal(m["ray"][2][l] > 0, m["ray"][1][lat.opp[l]] & (m["ray"][0][0] == 3) & (m["ray"][2][l] > 0))


def test_aabb_close_fills_the_shell(masks):
    lat, verts, m = masks
    shell = m["aabb"][0][0] == mb.BC_SOLID
    closed = m["close"][0][0] == mb.BC_SOLID
    assert (closed | ~shell).all()  # closing is extensive: the shell stays solid
    assert closed[7, 6, 6] and not shell[7, 6, 6]  # the cavity inside the shell is filled
    bnd = m["close"][0][0] == 3
    assert bnd.any() and not (bnd & closed).any()
    d = m["close"][2]
    assert d[d != 0].min() > -0.5 and d.max() <= 1.0
    # close_voxels = 0 would be the plain shell: the solid mask helper agrees with the AABB masker
    assert np.array_equal(mb.aabb_close_solid(SHAPE, verts, 0), shell)


@pytest.mark.parametrize("kind", mb.HYBRID_KINDS)
def test_hybrid_bc_keeps_the_fluid_at_rest(kind):
    """A no-slip HybridBC around a body in a fluid at rest is a fixed point: f = w stays f = w (interpolated bounce-back of
    equal populations, zero non-equilibrium part), with and without wall distances."""
    lat = orc.Lattice("D3Q19")
    shape = (10, 9, 8)
    x, y, z = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    body = np.array(np.where((x - 4.5) ** 2 + (y - 4) ** 2 + (z - 3.5) ** 2 < 2.2**2))
    for dist in (None, 0.3):
        bc = mb.HybridBC(kind, 1, body)
        bm, mm = orc.build_masks(shape, lat, [bc])
        if dist is not None:
            bc.distances = np.full((lat.q,) + shape, dist, np.float32)
        f0 = orc.initialize_eq(shape, lat)
        out = mb.run(f0, bm, mm, [bc], 1.4, lat, 3)
        assert np.abs(out - f0).max() < 1e-6  # rounding only


def test_hybrid_interpolated_bounceback_at_half_way_is_the_halfway_wall():
    """weight 1/2 ... : ((1 - w) f_post[opp] + w (f_pre[l] + f_pre[opp])) / (1 + w) is NOT f_pre[opp] in general, but without
    distances the interpolated bounce-back IS the halfway rule: check the pre-regularisation populations through Grad's
    variant, which only rewrites the missing directions."""
    lat = orc.Lattice("D3Q19")
    shape = (8, 8, 8)
    body = np.array([[4], [4], [4]])
    hy = mb.HybridBC(mb.KIND_HYBRID_BB_GRADS, 1, body)
    hw = orc.BC(orc.KIND_HALFWAY_BB, 1, body)
    bm, mm = orc.build_masks(shape, lat, [hy])
    bm2, mm2 = orc.build_masks(shape, lat, [hw])
    assert np.array_equal(bm, bm2) and np.array_equal(mm, mm2)
    f = orc.perturbed_init(shape, lat, seed=4)
    T = np.float32
    F0 = f.astype(T)
    post = orc.stream(F0, lat)
    a = mb.apply_hybrid(hy, F0, post, bm, mm, lat, "FP32FP32")
    b = orc.apply_bc(hw, F0, post, bm, mm, lat, "FP32FP32")
    known = ~mm.astype(bool)
    assert np.array_equal(a[known], b[known])  # non-missing populations untouched by Grad's variant
    assert not np.array_equal(a, b)  # the missing ones are re-expressed through the moments
    # moments of the halfway-wall state are what Grad's approximation is built from: density is preserved to rounding
    assert np.abs(a.sum(axis=0) - b.sum(axis=0)).max() < 5e-3
