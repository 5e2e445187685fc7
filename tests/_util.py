"""Shared helpers for the parity tests: build the same case for the HIP backend
(xlb_amd, through the C ABI) and for the oracle, and compare."""

import os

import numpy as np

import xlb_amd
from oracle import xlb_numpy as orc
from xlb_amd import ComputeBackend, PrecisionPolicy
from xlb_amd.grid import grid_factory
from xlb_amd.operator.boundary_condition import EquilibriumBC, FullwayBounceBackBC, HalfwayBounceBackBC
from xlb_amd.operator.macroscopic import Macroscopic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VS = {"D2Q9": xlb_amd.velocity_set.D2Q9, "D3Q19": xlb_amd.velocity_set.D3Q19, "D3Q27": xlb_amd.velocity_set.D3Q27}


def init_hip(lattice, policy="FP32FP32"):
    pp = PrecisionPolicy[policy]
    vs = VS[lattice](precision_policy=pp, compute_backend=ComputeBackend.HIP)
    xlb_amd.init(velocity_set=vs, default_backend=ComputeBackend.HIP, default_precision_policy=pp)
    return vs, pp


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cavity_indices(grid, dim):
    """The index bookkeeping of the reference drivers (lid_driven_cavity_2d.py:43-49,
    mlups_3d.py:193-199)."""
    box = grid.bounding_box_indices()
    box_ne = grid.bounding_box_indices(remove_edges=True)
    lid = box_ne["top"]
    faces = ["bottom", "left", "right"] + (["front", "back"] if dim == 3 else [])
    walls = [sum((box[f][i] for f in faces), []) for i in range(dim)]
    walls = np.unique(np.array(walls), axis=-1).tolist()
    return lid, walls


def hip_cavity_2d(n, u_lid=0.05):
    """[walls, lid] with lid constructed first (lid_driven_cavity_2d.py:51-55)."""
    vs, pp = init_hip("D2Q9")
    grid = grid_factory((n, n))
    lid, walls = cavity_indices(grid, 2)
    bc_top = EquilibriumBC(rho=1.0, u=(u_lid, 0.0), indices=lid)
    bc_walls = HalfwayBounceBackBC(indices=walls)
    bcs = [bc_walls, bc_top]
    lat = orc.Lattice("D2Q9")
    obcs = [orc.BC(orc.KIND_HALFWAY_BB, bc_walls.id, walls), orc.BC(orc.KIND_EQUILIBRIUM, bc_top.id, lid, rho=1.0, u=(u_lid, 0.0))]
    return grid, bcs, lat, obcs


def hip_cavity_3d(shape, walls_cls, lattice="D3Q19", policy="FP32FP32", u_lid=0.02, backend_config=None):
    """[lid, walls] (mlups_3d.py:201-204)."""
    vs, pp = init_hip(lattice, policy)
    grid = grid_factory(shape, backend_config=backend_config)
    lid, walls = cavity_indices(grid, 3)
    bc_lid = EquilibriumBC(rho=1.0, u=(u_lid, 0.0, 0.0), indices=lid)
    bc_walls = walls_cls(indices=walls)
    kind = orc.KIND_FULLWAY_BB if walls_cls is FullwayBounceBackBC else orc.KIND_HALFWAY_BB
    lat = orc.Lattice(lattice)
    obcs = [orc.BC(orc.KIND_EQUILIBRIUM, bc_lid.id, lid, rho=1.0, u=(u_lid, 0.0, 0.0)), orc.BC(kind, bc_walls.id, walls)]
    return grid, [bc_lid, bc_walls], lat, obcs


def remap_ids(mask, mapping):
    out = mask.copy()
    for src, dst in mapping.items():
        out[mask == src] = dst
    return out


def hip_macroscopic(f, grid, vs, pp):
    macro = Macroscopic(velocity_set=vs, precision_policy=pp, compute_backend=ComputeBackend.HIP)
    rho = grid.create_field(1, dtype=pp.compute_precision)
    u = grid.create_field(vs.d, dtype=pp.compute_precision)
    macro(f, rho, u)
    return rho.numpy(), u.numpy()


def max_ulp_diff(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    it = {2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
    ia, ib = a.view(it).astype(np.int64), b.view(it).astype(np.int64)
    sign = np.int64(1) << (8 * a.dtype.itemsize - 1)
    ia = np.where(ia < 0, -(ia + sign) - 0, ia)  # map sign-magnitude to a monotone integer line
    ib = np.where(ib < 0, -(ib + sign) - 0, ib)
    return int(np.abs(ia - ib).max())


def icosphere(center, radius, subdivisions=2):
    """triangle soup (3 n, 3) of a sphere, counter-clockwise seen from outside, built procedurally (no STL reader here)"""
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
