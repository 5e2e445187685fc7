"""xlb_amd.utils.voxelize_stl on the GPU: the STL reader + this backend's WINDING / AABB voxelisers (reference utils.py:248-283 goes
through trimesh's voxeliser)."""

import numpy as np
import pytest

from xlb_amd.utils import save_stl, voxelize_stl

from _util import icosphere, init_hip

pytestmark = pytest.mark.gpu


def test_voxelize_stl_sphere(tmp_path):
    init_hip("D3Q19")
    r = 0.37  # physical units
    p = tmp_path / "sphere.stl"
    save_stl(str(p), icosphere((1.0, 2.0, 3.0), r, 3))
    vox, pitch = voxelize_stl(str(p), length_lbm_unit=24)  # the largest extent (the diameter) spans 24 voxels
    assert np.isclose(pitch, 2 * r / 24, rtol=1e-3) and vox.pitch == pitch
    m = vox.matrix
    assert m.dtype == bool and all(24 <= n <= 24 + 6 for n in m.shape)
    # filled volume: the ball's, plus the half-voxel shell the surface-overlap test adds
    vol = m.sum() * pitch**3
    ball = 4.0 / 3.0 * np.pi * r**3
    assert 0.98 * ball < vol < 1.25 * ball
    # centred on the sphere's centre, symmetric under the three axis flips to within the discretisation
    c = vox.points.mean(axis=0)
    assert np.allclose(c, (1.0, 2.0, 3.0), atol=pitch)
    with pytest.raises(ValueError):
        voxelize_stl(str(p))
    vox2, pitch2 = voxelize_stl(str(p), pitch=2 * pitch)
    assert pitch2 == 2 * pitch and abs(vox2.matrix.sum() * pitch2**3 - ball) < 0.5 * ball
