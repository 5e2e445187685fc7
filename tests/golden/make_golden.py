"""Generates the committed golden fixtures from the NumPy oracle (oracle/xlb_numpy.py).

The reference ships no golden vectors and cannot be imported here (it hard-imports jax and
warp, neither installed), so these vectors are produced by the oracle — which is pinned by
the reference's own known-answer tests (tests/test_oracle_reference_pins.py).  Fixtures are
data only: inputs are re-creatable from the seeds/configs stored next to the outputs.

    python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import xlb_numpy as orc  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def macro(f, lat, policy):
    rho, u = orc.macroscopic(f.astype(orc.compute_dtype(policy)), lat)
    return rho, u


def cavity_2d(n, steps):
    lat, shape, bcs = orc.cavity_2d(n)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    omega = 1.0 / (3.0 * (0.05 * (n - 1) / 200.0) + 0.5)  # examples/cfd/lid_driven_cavity_2d.py:109-113
    f = orc.initialize_eq(shape, lat)
    out = {"bc_mask": bc_mask, "missing_mask": missing.astype(np.uint8), "omega": np.float64(omega), "steps": np.array(steps)}
    done = 0
    for s in steps:
        f = orc.run(f, bc_mask, missing, bcs, omega, lat, s - done)
        done = s
        rho, u = macro(f, lat, "FP32FP32")
        out[f"rho_{s}"] = rho
        out[f"u_{s}"] = u
        if n <= 16:
            out[f"f_{s}"] = f
    save(f"d2q9_cavity_{n}", **out)


def d3q19_periodic(n=16, steps=20):
    lat = orc.Lattice("D3Q19")
    shape = (n, n, n)
    bm = np.zeros((1,) + shape, np.uint8)
    mm = np.zeros((lat.q,) + shape, bool)
    out = {"seed": np.int64(0)}
    for omega in (1.0, 1.7):
        f = orc.perturbed_init(shape, lat, seed=0)
        f = orc.run(f, bm, mm, [], omega, lat, steps)
        out[f"f_omega{omega}"] = f
    save(f"d3q19_periodic_{n}", steps=np.int64(steps), **out)


def d3q19_cavity(n=16, steps=30):
    out = {}
    for kind, tag in ((orc.KIND_FULLWAY_BB, "fullway"), (orc.KIND_HALFWAY_BB, "halfway")):
        lat, shape, bcs = orc.cavity_3d(n, kind)
        bc_mask, missing = orc.build_masks(shape, lat, bcs)
        f = orc.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, 1.0, lat, steps)
        out[f"f_{tag}"] = f
        out["bc_mask"] = bc_mask
        out["missing_mask"] = np.packbits(missing.astype(np.uint8), axis=0)
    save(f"d3q19_cavity_{n}", steps=np.int64(steps), **out)


def d3q27_kbc(n=12, steps=10):
    lat = orc.Lattice("D3Q27")
    shape = (n, n, n)
    bm = np.zeros((1,) + shape, np.uint8)
    mm = np.zeros((lat.q,) + shape, bool)
    out = {}
    for policy in ("FP64FP32", "FP32FP32", "FP64FP64"):
        f = orc.perturbed_init(shape, lat, policy, seed=0)
        out[f"f_{policy}"] = orc.run(f, bm, mm, [], 1.9, lat, steps, policy, "KBC")
    save(f"d3q27_kbc_{n}", steps=np.int64(steps), omega=np.float64(1.9), **out)


def d2q9_kbc(n=16, steps=20):
    lat, shape, bcs = orc.cavity_2d(n)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    f = orc.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, 1.6, lat, steps, "FP32FP32", "KBC")
    save(f"d2q9_kbc_cavity_{n}", steps=np.int64(steps), omega=np.float64(1.6), f=f)


def sphere_channel(steps=30):
    """flow past a sphere: populations, vorticity magnitude / Q of the velocity field and
    the momentum-exchange force on the sphere"""
    shape = (28, 14, 14)
    lat, bcs, _ = orc.sphere_channel(shape)
    bc_mask, missing = orc.build_masks(shape, lat, bcs)
    with np.errstate(all="ignore"):
        f = orc.run(orc.initialize_eq(shape, lat), bc_mask, missing, bcs, 1.5, lat, steps)
    rho, u = macro(f, lat, "FP32FP32")
    zero = np.zeros((3,) + shape, np.float32)
    vort, mag = orc.vorticity(u, bc_mask, zero, zero[:1])
    _, q = orc.q_criterion(u, bc_mask, zero[:1], zero[:1])
    force = orc.momentum_transfer(f, bcs[3], bc_mask, missing, lat)
    save("d3q19_sphere_channel", steps=np.int64(steps), omega=np.float64(1.5), f=f, bc_mask=bc_mask, vorticity_magnitude=mag, q=q, force=force)


if __name__ == "__main__":
    cavity_2d(16, [1, 10, 100])
    cavity_2d(128, [1, 10, 100, 1000])
    d3q19_periodic()
    d3q19_cavity()
    d3q27_kbc()
    d2q9_kbc()
    sphere_channel()
