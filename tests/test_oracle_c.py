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
