"""The oracle against every known-answer property the reference's own tests hold
for the hot path (pins 1-7).  CPU only."""

import numpy as np
import pytest

from oracle import xlb_numpy as orc

LATTICES = [("D2Q9", (100, 100)), ("D3Q19", (50, 50, 50)), ("D3Q27", (50, 50, 50))]
SMALL = [("D2Q9", (50, 50)), ("D3Q19", (20, 20, 20)), ("D3Q27", (20, 20, 20))]


@pytest.mark.parametrize("name,shape", LATTICES)
def test_pin1_equilibrium_rest_state(name, shape):
    # reference tests/kernels/equilibrium/test_equilibrium_jax.py:43-48
    lat = orc.Lattice(name)
    rho = np.ones((1,) + shape, np.float32)
    u = np.zeros((lat.d,) + shape, np.float32)
    feq = orc.equilibrium(rho, u, lat, np.float32)
    assert np.allclose(np.sum(feq, axis=0), 1.0)
    for i in range(lat.q):
        assert np.allclose(feq[i], lat.w[i])


@pytest.mark.parametrize("name,shape", SMALL)
@pytest.mark.parametrize("rho0,u0", [(1.0, 0.0), (1.1, 1.0), (1.1, 2.0)])
def test_pin2_macroscopic_roundtrip(name, shape, rho0, u0):
    # reference tests/kernels/macroscopic/test_macroscopic_jax.py:23-50, _warp.py:23-49
    lat = orc.Lattice(name)
    rho = np.full((1,) + shape, rho0, np.float32)
    u = np.full((lat.d,) + shape, u0, np.float32)
    f = orc.equilibrium(rho, u, lat, np.float32)
    r, v = orc.macroscopic(f, lat)
    assert np.allclose(r, rho0)
    assert np.allclose(v, u0, atol=1e-6)


@pytest.mark.parametrize("name,shape", LATTICES)
@pytest.mark.parametrize("omega", [0.6, 1.0])
def test_pin3_bgk(name, shape, omega):
    # reference tests/kernels/collision/test_bgk_collision_jax.py:20-50
    lat = orc.Lattice(name)
    rho = np.ones((1,) + shape, np.float32)
    u = np.zeros((lat.d,) + shape, np.float32)
    feq = orc.equilibrium(rho, u, lat, np.float32)
    f = np.zeros_like(feq)
    out = orc.bgk(f, feq, omega)
    assert np.allclose(out, f - omega * (f - feq))
    assert np.allclose(out, omega * feq, atol=1e-7)


@pytest.mark.parametrize("name,shape", SMALL)
def test_pin4_stream_is_roll(name, shape):
    # reference tests/kernels/stream/test_stream_jax.py:38-65
    lat = orc.Lattice(name)
    f = np.zeros((lat.q,) + shape, np.float32)
    f[(slice(None), 1) + (slice(None),) * (lat.d - 1)] = 1.0  # one-hot plane
    rng = np.random.default_rng(3)
    f += rng.random(f.shape, dtype=np.float32)
    out = orc.stream(f, lat)
    for i in range(lat.q):
        exp = np.roll(f[i], tuple(lat.c[:, i]), axis=tuple(range(lat.d)))
        assert np.array_equal(out[i], exp)


def _sphere_indices(shape):
    n = shape[0]
    r = n // 4
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    d2 = sum((g - n // 2) ** 2 for g in grids)
    return [a.tolist() for a in np.where(d2 < r**2)]


@pytest.mark.parametrize("name,shape", SMALL)
def test_pin5_masker_ids(name, shape):
    # reference tests/boundary_conditions/mask/test_bc_indices_masker_jax.py:43-79
    lat = orc.Lattice(name)
    idx = _sphere_indices(shape)
    bc = orc.BC(orc.KIND_FULLWAY_BB, 5, idx)
    bc_mask, missing = orc.build_masks(shape, lat, [bc])
    assert bc_mask.dtype == np.uint8 and missing.dtype == bool
    assert bc_mask.shape == (1,) + shape and missing.shape == (lat.q,) + shape
    assert np.all(bc_mask[(0,) + tuple(np.array(idx))] == 5)
    tmp = bc_mask.copy()
    tmp[(0,) + tuple(np.array(idx))] = 0
    assert np.all(tmp == 0)


@pytest.mark.parametrize("name,shape", SMALL)
def test_pin6_equilibrium_bc(name, shape):
    # reference tests/boundary_conditions/bc_equilibrium/test_bc_equilibrium_jax.py:55-90
    lat = orc.Lattice(name)
    idx = _sphere_indices(shape)
    bc = orc.BC(orc.KIND_EQUILIBRIUM, 1, idx, rho=1.0, u=(0.0,) * lat.d)
    bc_mask, missing = orc.build_masks(shape, lat, [bc])
    f_pre = np.zeros((lat.q,) + shape, np.float32)
    f_post = np.full((lat.q,) + shape, 2.0, np.float32)
    out = orc.apply_bc(bc, f_pre, f_post, bc_mask, missing, lat, "FP32FP32")
    inside = bc_mask[0] == 1
    for i in range(lat.q):
        assert np.allclose(out[i][inside], lat.w[i])
        assert np.allclose(out[i][~inside], 2.0)


@pytest.mark.parametrize("name,shape", SMALL)
def test_pin7_fullway_bb(name, shape):
    # reference tests/boundary_conditions/bc_fullway_bounce_back/test_bc_fullway_bounce_back_jax.py:55-90
    lat = orc.Lattice(name)
    idx = _sphere_indices(shape)
    bc = orc.BC(orc.KIND_FULLWAY_BB, 1, idx)
    bc_mask, missing = orc.build_masks(shape, lat, [bc])
    rng = np.random.default_rng(0)
    f_pre = rng.random((lat.q,) + shape, dtype=np.float32)
    f_post = rng.random((lat.q,) + shape, dtype=np.float32)
    out = orc.apply_bc(bc, f_pre, f_post, bc_mask, missing, lat, "FP32FP32")
    inside = bc_mask[0] == 1
    for i in range(lat.q):
        assert np.array_equal(out[i][~inside], f_post[i][~inside])
        assert np.array_equal(out[i][inside], f_pre[lat.opp[i]][inside])


def test_lattice_tables_match_survey_appendix_a():
    # opposite tables and face sets as derived in SURVEY.md appendix A
    l19 = orc.Lattice("D3Q19")
    assert l19.opp.tolist() == [0, 2, 1, 6, 8, 7, 3, 5, 4, 14, 16, 15, 18, 17, 9, 11, 10, 13, 12]
    assert l19.right.tolist() == [14, 15, 16, 17, 18] and l19.left.tolist() == [9, 10, 11, 12, 13]
    l9 = orc.Lattice("D2Q9")
    assert l9.opp.tolist() == [0, 2, 1, 6, 5, 4, 3, 8, 7]
    l27 = orc.Lattice("D3Q27")
    assert l27.opp.tolist() == [0, 2, 1, 6, 8, 7, 3, 5, 4, 18, 20, 19, 24, 26, 25, 21, 23, 22, 9, 11, 10, 15, 17, 16, 12, 14, 13]
    for lat in (l9, l19, l27):
        assert abs(lat.w.sum() - 1.0) < 1e-15


def test_masker_counts_cavity_2d():
    # SURVEY.md appendix B.2 sanity counts for the 128x128 cavity
    lat, shape, bcs = orc.cavity_2d(128)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    assert int((bc_mask == 2).sum()) == 382 and int((bc_mask == 1).sum()) == 126
    assert int(missing.sum()) == 1532


def test_kbc_reduces_to_bgk_form_when_delta_h_zero():
    # physics invariant: with gamma = 2 (fneq purely "shear" so delta_h = 0) KBC = BGK
    lat = orc.Lattice("D2Q9")
    rng = np.random.default_rng(1)
    shape = (8, 8)
    rho = (1 + 0.01 * rng.uniform(-1, 1, (1,) + shape)).astype(np.float64)
    u = (0.01 * rng.uniform(-1, 1, (2,) + shape)).astype(np.float64)
    feq = orc.equilibrium(rho, u, lat, np.float64)
    out = orc.kbc(feq.copy(), feq, 1.7, lat)
    assert np.allclose(out, feq, atol=1e-15)  # fneq = 0 is a fixed point


def test_mass_conservation_periodic():
    lat = orc.Lattice("D3Q19")
    shape = (8, 8, 8)
    f = orc.perturbed_init(shape, lat, "FP64FP64")
    bm = np.zeros((1,) + shape, np.uint8)
    mm = np.zeros((lat.q,) + shape, bool)
    m0 = f.sum()
    f = orc.run(f, bm, mm, [], 1.7, lat, 10, "FP64FP64")
    assert abs(f.sum() - m0) < 1e-10
