"""The C restatement (oracle/lbm_ref.c) against the NumPy oracle and the golden fixtures:
two independently written CPU restatements of the reference step must agree bit for bit."""

import os

import numpy as np
import pytest

from oracle import lbm_ref, xlb_numpy as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.skipif(not os.path.exists(lbm_ref.LIB_PATH), reason="oracle/liblbmref.so not built (run `make oracle`)")


def test_c_oracle_cavity_2d_golden():
    g = np.load(os.path.join(GOLDEN, "d2q9_cavity_16.npz"))
    lat, shape, bcs = orc.cavity_2d(16)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    f = lbm_ref.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, float(g["omega"]), lat, 100)
    assert np.array_equal(f, g["f_100"])


@pytest.mark.parametrize("kind,tag", [(orc.KIND_FULLWAY_BB, "fullway"), (orc.KIND_HALFWAY_BB, "halfway")])
def test_c_oracle_cavity_3d_golden(kind, tag):
    g = np.load(os.path.join(GOLDEN, "d3q19_cavity_16.npz"))
    lat, shape, bcs = orc.cavity_3d(16, kind)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    f = lbm_ref.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, 1.0, lat, int(g["steps"]), threads=4)
    assert np.array_equal(f, g[f"f_{tag}"])


@pytest.mark.parametrize("policy", ["FP32FP32", "FP64FP64"])
def test_c_oracle_kbc_golden(policy):
    g = np.load(os.path.join(GOLDEN, "d3q27_kbc_12.npz"))
    lat = orc.Lattice("D3Q27")
    f0 = orc.perturbed_init((12, 12, 12), lat, policy, seed=0)
    f = lbm_ref.run(f0, None, None, [], float(g["omega"]), lat, int(g["steps"]), policy, "KBC", threads=2)
    assert np.array_equal(f, g[f"f_{policy}"])


@pytest.mark.parametrize("lattice,shape,collision", [("D3Q19", (9, 7, 11), "BGK"), ("D3Q27", (6, 5, 7), "KBC"), ("D2Q9", (13, 17), "KBC")])
def test_c_oracle_all_bcs_random(lattice, shape, collision):
    lat = orc.Lattice(lattice)
    box = orc.bounding_box_indices(shape, remove_edges=True)
    u0 = (0.02,) + (0.0,) * (lat.d - 1)
    bcs = [
        orc.BC(orc.KIND_HALFWAY_BB, 7, box["bottom"], u_wall=(0.0,) * (lat.d - 1) + (0.0,) if False else tuple(0.01 * (i + 1) for i in range(lat.d))),
        orc.BC(orc.KIND_EQUILIBRIUM, 3, box["top"], rho=1.0, u=u0),
        orc.BC(orc.KIND_FULLWAY_BB, 9, box["left"]),
        orc.BC(orc.KIND_DO_NOTHING, 4, box["right"]),
    ]
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    f0 = orc.perturbed_init(shape, lat, seed=9)
    exp = orc.run(f0, bc_mask, missing, bcs, 1.6, lat, 6, "FP32FP32", collision)
    out = lbm_ref.run(f0, bc_mask, missing, bcs, 1.6, lat, 6, "FP32FP32", collision, threads=3)
    assert np.array_equal(out, exp)


def test_oracle_reproduces_committed_goldens():
    """Regression pin: the NumPy oracle still produces the committed fixtures."""
    g = np.load(os.path.join(GOLDEN, "d3q19_periodic_16.npz"))
    lat = orc.Lattice("D3Q19")
    shape = (16, 16, 16)
    bm = np.zeros((1,) + shape, np.uint8)
    mm = np.zeros((lat.q,) + shape, bool)
    f = orc.run(orc.perturbed_init(shape, lat, seed=0), bm, mm, [], 1.7, lat, int(g["steps"]))
    assert np.array_equal(f, g["f_omega1.7"])
    g2 = np.load(os.path.join(GOLDEN, "d2q9_cavity_16.npz"))
    lat2, shape2, bcs2 = orc.cavity_2d(16)
    bc_mask, missing = orc.build_masks(shape2, lat2, bcs2)
    assert np.array_equal(bc_mask, g2["bc_mask"]) and np.array_equal(missing.astype(np.uint8), g2["missing_mask"])


def test_outflow_auxiliary_data_matches_per_cell_restatement():
    """oracle.assemble_auxiliary_data is roll-based like the reference (bc_extrapolation_outflow.py:104-134); check it
    against the per-cell statement the HIP kernel implements: at an outflow cell b with outward normal n, every
    direction l whose opposite is missing gets cs * ps[opp l](b - n) + (1 - cs) * ps[opp l](b), ps = post-stream."""
    from oracle import xlb_numpy as orc

    lat = orc.Lattice("D3Q19")
    shape = (7, 5, 6)
    box = orc.bounding_box_indices(shape, remove_edges=True)
    bc = orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, 3, box["top"])
    assert bc.normal.tolist() == [0, 0, 1]
    low = orc.BC(orc.KIND_EXTRAPOLATION_OUTFLOW, 4, box["left"])
    assert low.normal.tolist() == [-1, 0, 0]
    bm, mm = orc.build_masks(shape, lat, [bc])
    rng = np.random.default_rng(0)
    ps = rng.random((19,) + shape).astype(np.float32)
    pc = rng.random((19,) + shape).astype(np.float32)
    got = orc.assemble_auxiliary_data(bc, ps, pc, bm, mm, lat)
    exp = pc.copy()
    cs = np.float32(1.0) / np.sqrt(np.float32(3.0))
    for x, y, z in zip(*[np.asarray(v) for v in box["top"]]):
        for l in range(19):
            o = lat.opp[l]
            if mm[o, x, y, z]:
                exp[l, x, y, z] = cs * ps[o, x, y, z - 1] + (np.float32(1.0) - cs) * ps[o, x, y, z]
    assert np.array_equal(got, exp) and not np.array_equal(got, pc)
    # every other BC kind leaves the post-collision populations alone (boundary_condition.py:138-144)
    wall = orc.BC(orc.KIND_HALFWAY_BB, 5, box["bottom"])
    assert orc.assemble_auxiliary_data(wall, ps, pc, bm, mm, lat) is pc


def test_sphere_channel_golden_pins_the_oracle():
    """The widened rows (profile inlet, extrapolation outflow, interior halfway sphere, vorticity / Q, momentum exchange)
    have no reference test: the committed vectors pin the oracle's restatement of them against later edits."""
    from oracle import xlb_numpy as orc

    g = np.load(os.path.join(GOLDEN, "d3q19_sphere_channel.npz"))
    shape = (28, 14, 14)
    lat, bcs, prof = orc.sphere_channel(shape)
    assert prof.shape == (3, 14, 14) and [b.id for b in bcs] == [1, 2, 3, 4] and bcs[2].normal.tolist() == [1, 0, 0]
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    assert np.array_equal(bc_mask, g["bc_mask"])
    with np.errstate(all="ignore"):
        f = orc.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, float(g["omega"]), lat, int(g["steps"]))
    assert np.array_equal(f, g["f"])
    rho, u = orc.macroscopic(f, lat)
    zero = np.zeros((3,) + shape, np.float32)
    assert np.array_equal(orc.vorticity(u, bc_mask, zero, zero[:1])[1], g["vorticity_magnitude"])
    assert np.array_equal(orc.q_criterion(u, bc_mask, zero[:1], zero[:1])[1], g["q"])
    assert np.array_equal(orc.momentum_transfer(f, bcs[3], bc_mask, missing, lat), g["force"])
    assert g["force"][0] > 0  # drag


def test_vectorised_cpu_port_matches_oracle():
    """oracle/lbm_cpu_fast.cpp (bench.py's optimised CPU baseline, compiled on this host): same step as the NumPy oracle to
    rounding (contraction and flush-to-zero allowed there: not bit-exact), every basic BC kind, D3Q19 and D3Q27."""
    lib = lbm_ref.build_cpu_port()
    if lib is None:
        pytest.skip("g++ not available")
    for kind in (orc.KIND_HALFWAY_BB, orc.KIND_FULLWAY_BB):
        lat, shape, bcs = orc.cavity_3d(14, kind)
        bm, mm = orc.build_masks(shape, lat, bcs)
        f = orc.perturbed_init(shape, lat, seed=3)
        exp = orc.run(f, bm, mm, bcs, 1.3, lat, 5)
        got = lbm_ref.run_cpu_port(lib, f, bm, mm, bcs, 1.3, lat, 5, threads=2)
        assert np.abs(got - exp).max() <= 1e-6
    lat = orc.Lattice("D3Q27")
    shape = (9, 7, 11)
    f = orc.perturbed_init(shape, lat, seed=1)
    bm, mm = orc.build_masks(shape, lat, [])
    assert np.abs(lbm_ref.run_cpu_port(lib, f, None, None, [], 1.7, lat, 4) - orc.run(f, bm, mm, [], 1.7, lat, 4)).max() <= 1e-6
