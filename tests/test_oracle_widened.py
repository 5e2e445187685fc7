"""CPU checks of the oracle's restatements of the widened rows through properties that do not
depend on any implementation: the GPU tests compare the HIP kernels with these functions bit for bit, so they must be right
on their own."""

import numpy as np

from oracle import xlb_numpy as orc


def icosphere(center, radius, subdivisions=1):
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


def test_aabb_voxelisation_gives_a_closed_shell_at_the_right_radius():
    from scipy import ndimage

    lat = orc.Lattice("D3Q19")
    shape = (14, 13, 12)
    c, r = np.array([6.3, 6.6, 5.9]), 3.1
    bc, mm = orc.mesh_mask_aabb(shape, lat, 7, icosphere(c, r), np.zeros((1,) + shape, np.uint8), np.zeros((19,) + shape, bool))
    solid = bc[0] == orc.BC_SOLID
    lab, _ = ndimage.label(~solid)
    assert lab[6, 6, 5] != lab[0, 0, 0]  # inside and outside are separated
    # every solid voxel's box comes within half a diagonal of the sphere surface
    x, y, z = np.where(solid)
    dist = np.abs(np.linalg.norm(np.stack([x, y, z], 1) + 0.5 - c, axis=1) - r)
    assert dist.max() <= 0.5 * 3**0.5 + 0.15  # (+ the chord error of the coarse icosphere)
    # boundary voxels are exactly the non-solid voxels with a solid lattice neighbour, and miss exactly those links
    for l in range(1, 19):
        cx, cy, cz = (int(v) for v in lat.c[:, l])
        nb = np.zeros(shape, bool)
        src = solid[max(cx, 0) : shape[0] + min(cx, 0), max(cy, 0) : shape[1] + min(cy, 0), max(cz, 0) : shape[2] + min(cz, 0)]
        nb[max(-cx, 0) : shape[0] + min(-cx, 0), max(-cy, 0) : shape[1] + min(-cy, 0), max(-cz, 0) : shape[2] + min(-cz, 0)] = src
        assert np.array_equal(mm[lat.opp[l]] & ~solid, nb & ~solid)
    assert np.array_equal(bc[0] == 7, mm.any(axis=0) & ~solid)


def test_ray_voxelisation_links_are_symmetric_and_straddle_the_surface():
    lat = orc.Lattice("D3Q19")
    shape = (12, 12, 12)
    c, r = np.array([5.8, 6.1, 5.7]), 2.7
    bc, mm = orc.mesh_mask_ray(shape, lat, 3, icosphere(c, r), np.zeros((1,) + shape, np.uint8), np.zeros((19,) + shape, bool))
    assert not (bc[0] == orc.BC_SOLID).any()
    x, y, z = np.where(bc[0] == 3)
    rad = np.linalg.norm(np.stack([x, y, z], 1) + 0.5 - c, axis=1)
    for l in range(1, 19):
        sel = mm[lat.opp[l], x, y, z]
        xn, yn, zn = x[sel] + lat.c[0, l], y[sel] + lat.c[1, l], z[sel] + lat.c[2, l]
        assert np.all(mm[l, xn, yn, zn])  # the link is missing from both ends
        rn = np.linalg.norm(np.stack([xn, yn, zn], 1) + 0.5 - c, axis=1)
        # its ends lie on opposite sides of the sphere (up to the chord error of the coarse mesh)
        assert np.all((rad[sel] - r) * (rn - r) <= 0.2)


def test_vorticity_q_and_probe_on_analytic_fields():
    shape = (9, 8, 7)
    x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    om = np.array([0.3, -0.2, 0.5])
    u = np.stack([om[1] * z - om[2] * y, om[2] * x - om[0] * z, om[0] * y - om[1] * x])
    bm = np.zeros((1,) + shape, np.uint8)
    zero = np.zeros((3,) + shape)
    vort, mag = orc.vorticity(u, bm, zero, zero[:1])
    core = (slice(1, -1),) * 3
    assert np.allclose(vort[(slice(None),) + core], (2 * om)[:, None, None, None]) and np.allclose(mag[0][core], 2 * np.linalg.norm(om))
    _, q = orc.q_criterion(u, bm, zero[:1], zero[:1])
    assert np.allclose(q[0][core], om @ om) and np.all(q[0, 0] == 0)
    bm[0, 4, 4, 3] = 9  # a boundary cell: its six neighbours are skipped, it is not
    _, mag2 = orc.vorticity(u, bm, zero, zero[:1])
    assert mag2[0, 3, 4, 3] == 0 and mag2[0, 4, 4, 3] != 0
    lin = (0.5 * x - 0.25 * y + 2.0 * z + 1.0)[None]
    pts = np.array([[1.25, 2.5, 3.75], [0.0, 0.0, 0.0], [6.9, 5.1, 4.2]], np.float32)
    assert np.allclose(orc.grid_to_point(lin, pts), 0.5 * pts[:, 0] - 0.25 * pts[:, 1] + 2.0 * pts[:, 2] + 1.0, atol=1e-5)


def test_momentum_transfer_of_a_uniform_stream_past_a_block():
    """populations at rest equilibrium except a surplus moving along +x: the windward faces of a block feel a force along +x"""
    lat = orc.Lattice("D3Q19")
    shape = (10, 8, 8)
    block = np.array(np.where(np.ones((2, 2, 2), bool))) + np.array([[4], [3], [3]])
    bc = orc.BC(orc.KIND_HALFWAY_BB, 1, block)
    bm, mm = orc.build_masks(shape, lat, [bc])
    f = np.broadcast_to(lat.w.astype(np.float32).reshape(19, 1, 1, 1), (19,) + shape).copy()
    l_px = [l for l in range(19) if tuple(lat.c[:, l]) == (1, 0, 0)][0]
    f[l_px] += 0.01
    force = orc.momentum_transfer(f, bc, bm, mm, lat)
    assert force[0] > 0 and abs(force[1]) < 1e-6 and abs(force[2]) < 1e-6
