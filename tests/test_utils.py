"""xlb_amd.utils — the host-side helpers the reference's drivers import from xlb.utils (utils.py:28-537): output files, STL input,
geometry helpers, unit conversion.  CPU only (the STL voxeliser is in tests/test_gpu_utils.py)."""

import numpy as np
import pytest

from xlb_amd.utils import UnitConvertor, axangle2mat, downsample_field, load_stl, read_fields_vtk, rotate_geometry, save_fields_vtk, save_image, save_stl

from _util import icosphere


@pytest.mark.parametrize("shape", [(5, 7), (4, 6, 3)])
def test_save_fields_vtk_round_trip(tmp_path, shape):
    rng = np.random.default_rng(3)
    fields = {"rho": rng.random(shape).astype(np.float32), "u_x": rng.random(shape), "bc": rng.integers(0, 5, shape).astype(np.uint8)}
    path = save_fields_vtk(fields, 12, output_dir=str(tmp_path / "out"), prefix="flds")
    assert path.endswith("flds_0000012.vtk")  # utils.py:126-137: '<prefix>_<timestep:07d>.vtk'
    head = open(path, "rb").read(200).decode("latin1")
    pts = " ".join(str(n + 1) for n in shape) + (" 1" if len(shape) == 2 else "")
    assert "DATASET STRUCTURED_POINTS" in head and f"DIMENSIONS {pts}" in head  # cell-centred: one more point than cells per axis
    back = read_fields_vtk(path)
    assert set(back) == set(fields)
    for k, v in fields.items():
        assert back[k].shape == shape and np.array_equal(back[k], v.astype(back[k].dtype))
    with pytest.raises(AssertionError):
        save_fields_vtk({"a": np.zeros(shape), "b": np.zeros(shape[::-1])}, 0, output_dir=str(tmp_path))


def test_save_image_names_and_sizes(tmp_path):
    import matplotlib.image as mpimg

    scal = np.random.default_rng(1).random((12, 7))
    p1 = save_image(scal, timestep=5, prefix=str(tmp_path / "img"))
    assert p1.endswith("img_0005.png")  # utils.py:86-87: zero-filled to four digits
    assert mpimg.imread(p1).shape[:2] == (7, 12)  # the first axis runs to the right (fld.T)
    vec = np.random.default_rng(2).random((3, 6, 9))
    p2 = save_image(vec, prefix=str(tmp_path / "mag"))
    assert p2.endswith("mag.png") and mpimg.imread(p2).shape[:2] == (9, 6)
    with pytest.raises(ValueError):
        save_image(np.zeros((2, 3, 4, 5)), prefix=str(tmp_path / "bad"))


def test_axangle2mat_and_rotate_geometry():
    r = axangle2mat((0, 0, 2.0), np.pi / 2)
    assert np.allclose(r @ r.T, np.eye(3)) and np.isclose(np.linalg.det(r), 1.0)
    assert np.allclose(r @ np.array([1.0, 0, 0]), [0, 1, 0])  # a quarter turn about z takes x to y
    k = np.array([1.0, -2.0, 0.5])
    assert np.allclose(axangle2mat(k, 0.7) @ k, k)  # the axis is fixed
    assert np.allclose(axangle2mat(k / np.linalg.norm(k), 0.7, is_normalized=True), axangle2mat(k, 0.7))
    # voxel indices about the axis through `origin`: row vectors times R (utils.py:244), i.e. the inverse rotation of the points
    idx = ([3, 4, 5], [2, 2, 2], [1, 1, 1])
    out = rotate_geometry(idx, origin=(3, 2, 1), axis=(0, 0, 1), angle=np.pi / 2)
    assert [o.tolist() for o in out] == [[3, 3, 3], [2, 1, 0], [1, 1, 1]]
    assert out[0].dtype == np.int32


def test_stl_round_trip_binary_and_ascii(tmp_path):
    verts = icosphere((4.0, 5.0, 6.0), 2.5, 1)
    p = tmp_path / "s.stl"
    save_stl(str(p), verts)
    back = load_stl(str(p))
    assert back.dtype == np.float32 and back.shape == verts.shape and np.array_equal(back, verts.astype(np.float32))
    tri = back.reshape(-1, 3, 3)[:3]
    text = "solid t\n" + "".join(
        "facet normal 0 0 0\n outer loop\n" + "".join(f"  vertex {v[0]!r} {v[1]!r} {v[2]!r}\n" for v in t.tolist()) + " endloop\nendfacet\n" for t in tri
    ) + "endsolid t\n"
    pa = tmp_path / "a.stl"
    pa.write_text(text)
    assert np.allclose(load_stl(str(pa)), tri.reshape(-1, 3))
    (tmp_path / "junk.stl").write_bytes(b"not an stl")
    with pytest.raises(ValueError):
        load_stl(str(tmp_path / "junk.stl"))


def test_unit_convertor_identities():
    uc = UnitConvertor(velocity_lbm_unit=0.05, velocity_physical_unit=10.0, voxel_size_physical_unit=0.01)
    assert np.isclose(uc.time_step_physical, 0.01 * 0.05 / 10.0)  # dt = dx u_lbm / u_phys
    assert np.isclose(uc.velocity_to_lbm(10.0), 0.05) and np.isclose(uc.velocity_to_physical(0.05), 10.0)
    for to_l, to_p, x in [(uc.length_to_lbm, uc.length_to_physical, 0.37), (uc.time_to_lbm, uc.time_to_physical, 2.5), (uc.density_to_lbm, uc.density_to_physical, 1.1),
                          (uc.viscosity_to_lbm, uc.viscosity_to_physical, 1.5e-5)]:
        assert np.isclose(to_p(to_l(x)), x)
    # pressures: to_lbm gives the perturbation p' / (rho u_ref^2), to_physical takes a lattice pressure rho c_s^2 (1/3 at the reference state)
    assert np.isclose(uc.pressure_to_physical(uc.pressure_to_lbm(1.0e5) + 1.0 / 3.0), 1.0e5)
    assert np.isclose(uc.pressure_to_physical(1.0 / 3.0), 1.101325e5)  # rho c_s^2 at the reference state


def test_downsample_field_shapes_and_means():
    f = np.random.default_rng(5).random((16, 12, 3))
    assert downsample_field(f, 1) is f
    d = downsample_field(f, 2, method="bilinear")
    assert d.shape == (8, 6, 3)
    const = np.full((8, 8, 8, 2), 3.25)
    assert np.allclose(downsample_field(const, 4), 3.25) and downsample_field(const, 4).shape == (2, 2, 2, 2)
