"""Output, geometry and unit helpers with the reference's names and argument meaning (xlb/utils/utils.py), written against NumPy:

* ``save_image`` (utils.py:58-98) — PNG of a 2-D field or of the magnitude of a 3-component one, through matplotlib;
* ``save_fields_vtk`` (utils.py:101-153) — the reference goes through pyvista; here a legacy-VTK STRUCTURED_POINTS file with the fields
  as CELL_DATA is written directly (same file name, same cell-centred convention, readable by ParaView / pyvista); ``read_fields_vtk``
  reads such a file back;
* ``downsample_field`` (utils.py:28-55), ``rotate_geometry`` / ``axangle2mat`` (utils.py:219-327), ``UnitConvertor`` (utils.py:450-537);
* ``load_stl`` / ``save_stl`` — the reference loads meshes with trimesh; a binary / ASCII STL reader is all its drivers need
  (``mesh.vertices`` of an unprocessed load = the triangle soup the mesh BCs take);
* ``voxelize_stl`` (utils.py:248-283) — occupancy of a mesh on a grid of the given pitch, through this backend's WINDING voxeliser.
"""

import os
import struct
import sys
import time

import numpy as np


def _as_numpy(a):
    """Fields of this backend, NumPy arrays and anything with __array__."""
    return a.numpy() if hasattr(a, "numpy") and not isinstance(a, np.ndarray) else np.asarray(a)


# ---- images -----------------------------------------------------------------------------------------------------------------

def save_image(fld, timestep=None, prefix=None, **kwargs):
    """``<prefix or main script name>[_<timestep:04d>].png`` of a 2-D field; a (3, n0, n1) field is reduced to its magnitude.
    'nipy_spectral' colours, origin lower, first axis to the right (the reference's orientation)."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fld = _as_numpy(fld)
    if prefix is None:
        main = sys.modules.get("__main__")
        name = os.path.splitext(os.path.basename(getattr(main, "__file__", "xlb_amd")))[0]
    else:
        name = prefix
    if timestep is not None:
        name = f"{name}_{str(timestep).zfill(4)}"
    if fld.ndim > 3:
        raise ValueError("The input field should be 2D!")
    if fld.ndim == 3:
        fld = np.sqrt(fld[0] ** 2 + fld[1] ** 2 + fld[2] ** 2)
    kwargs.pop("cmap", None)
    plt.imsave(name + ".png", fld.T, cmap="nipy_spectral", origin="lower", **kwargs)
    return name + ".png"


# ---- VTK --------------------------------------------------------------------------------------------------------------------

def save_fields_vtk(fields, timestep, output_dir=".", prefix="fields"):
    """``{output_dir}/{prefix}_{timestep:07d}.vtk``: every entry of ``fields`` (name -> (nx, ny) or (nx, ny, nz) array, all of one
    shape) as a cell-centred scalar of a uniform grid with nx+1 x ny+1 (x nz+1) points.  Legacy VTK, binary (big-endian)."""
    names = list(fields.keys())
    if not names:
        raise ValueError("no fields to save")
    arrays = {k: _as_numpy(v) for k, v in fields.items()}
    shape = arrays[names[0]].shape
    for k in names:
        assert arrays[k].shape == shape, "All fields must have the same dimensions!"
    if len(shape) not in (2, 3):
        raise ValueError("fields must be 2-D or 3-D arrays")
    cells = tuple(shape) + ((1,) if len(shape) == 2 else ())
    points = tuple(n + 1 for n in shape) + ((1,) if len(shape) == 2 else ())
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"{prefix}_{timestep:07d}.vtk")
    t0 = time.time()
    with open(path, "wb") as fh:
        fh.write(b"# vtk DataFile Version 3.0\nxlb_amd fields\nBINARY\nDATASET STRUCTURED_POINTS\n")
        fh.write(f"DIMENSIONS {points[0]} {points[1]} {points[2]}\nORIGIN 0 0 0\nSPACING 1 1 1\n".encode())
        fh.write(f"CELL_DATA {int(np.prod(cells))}\n".encode())
        for k in names:
            a = arrays[k]
            if a.dtype.kind == "f":
                vtk_type, be = ("double", ">f8") if a.dtype.itemsize == 8 else ("float", ">f4")
            elif a.dtype.kind in "iub":
                vtk_type, be = "int", ">i4"
            else:
                raise ValueError(f"field {k}: unsupported dtype {a.dtype}")
            fh.write(f"SCALARS {k.replace(' ', '_')} {vtk_type} 1\nLOOKUP_TABLE default\n".encode())
            fh.write(np.asarray(a, order="C").flatten(order="F").astype(be).tobytes())  # x fastest, as VTK stores points
            fh.write(b"\n")
    print(f"Saved {path} in {time.time() - t0:.6f} seconds.")
    return path


def read_fields_vtk(path):
    """The fields of a file ``save_fields_vtk`` wrote: name -> array of the original shape."""
    with open(path, "rb") as fh:
        blob = fh.read()
    pos = 0

    def line():
        nonlocal pos
        end = blob.index(b"\n", pos)
        out = blob[pos:end].decode()
        pos = end + 1
        return out

    assert line().startswith("# vtk DataFile"), "not a legacy VTK file"
    line()
    assert line() == "BINARY" and line() == "DATASET STRUCTURED_POINTS"
    dims = [int(v) for v in line().split()[1:]]
    line()
    line()
    n = int(line().split()[1])
    cells = [max(d - 1, 1) for d in dims]
    shape = tuple(cells[:2]) if dims[2] == 1 else tuple(cells)
    out = {}
    while pos < len(blob):
        head = line()
        if not head:
            continue
        _, name, vtk_type, _ = head.split()
        line()
        be = {"double": ">f8", "float": ">f4", "int": ">i4"}[vtk_type]
        size = n * np.dtype(be).itemsize
        data = np.frombuffer(blob, dtype=be, count=n, offset=pos)
        pos += size + 1
        out[name] = data.astype(data.dtype.newbyteorder("=")).reshape(shape, order="F")
    return out


# ---- resampling / geometry ------------------------------------------------------------------------------------------------------

def downsample_field(field, factor, method="bicubic"):
    """A (..., ncomp) field resampled to ``dim // factor`` along every leading axis, component by component.
    ``method``: 'nearest', 'linear' / 'bilinear' / 'trilinear', 'bicubic' / 'cubic' (spline orders 0 / 1 / 3)."""
    field = _as_numpy(field)
    if factor == 1:
        return field
    from scipy import ndimage

    order = {"nearest": 0, "linear": 1, "bilinear": 1, "trilinear": 1, "bicubic": 3, "cubic": 3}[method]
    new_shape = tuple(d // factor for d in field.shape[:-1])
    zoom = [n / d for n, d in zip(new_shape, field.shape[:-1])]
    comps = [ndimage.zoom(field[..., i], zoom, order=order, mode="nearest", grid_mode=True)[tuple(slice(0, n) for n in new_shape)]
             for i in range(field.shape[-1])]
    return np.stack(comps, axis=-1)


def axangle2mat(axis, angle, is_normalized=False):
    """Rotation matrix (3, 3) of the rotation by ``angle`` (radians) about ``axis`` — Rodrigues' formula
    R = cos(a) I + sin(a) [k]x + (1 - cos(a)) k k^T."""
    k = np.asarray(axis, dtype=np.float64)
    if not is_normalized:
        k = k / np.linalg.norm(k)
    kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.cos(angle) * np.eye(3) + np.sin(angle) * kx + (1.0 - np.cos(angle)) * np.outer(k, k)


def rotate_geometry(indices, origin, axis, angle):
    """Voxel indices (3 sequences) rotated about the axis through ``origin``: ``(p - origin) @ R + origin``, rounded to the
    nearest voxel (the reference multiplies row vectors from the left, utils.py:244)."""
    p = np.asarray(indices, dtype=np.float64).T
    o = np.asarray(origin, dtype=np.float64)
    rotated = (p - o) @ axangle2mat(axis, angle) + o
    return tuple(np.rint(rotated).astype(np.int32).T)


# ---- STL -----------------------------------------------------------------------------------------------------------------------

def load_stl(filename):
    """Triangle soup (3 n, 3) float32 of a binary or ASCII STL file: three consecutive rows per facet, what ``mesh_vertices=`` takes
    (and what ``trimesh.load_mesh(f, process=False).vertices`` gives the reference's drivers)."""
    with open(filename, "rb") as fh:
        blob = fh.read()
    if len(blob) >= 84:
        (n,) = struct.unpack_from("<I", blob, 80)
        if len(blob) == 84 + 50 * n:  # binary: 80-byte header, count, 50 bytes per facet (normal, 3 vertices, attribute)
            rec = np.frombuffer(blob, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
            return np.ascontiguousarray(rec["v"].reshape(-1, 3), dtype=np.float32)
    verts = []
    for raw in blob.decode("ascii", errors="replace").splitlines():
        parts = raw.split()
        if len(parts) == 4 and parts[0] == "vertex":
            verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
    if not verts or len(verts) % 3:
        raise ValueError(f"{filename}: neither a binary nor an ASCII STL file")
    return np.asarray(verts, dtype=np.float32)


def save_stl(filename, vertices):
    """Binary STL of a triangle soup (3 n, 3); facet normals from the vertex order."""
    v = np.asarray(vertices, dtype=np.float32).reshape(-1, 3, 3)
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    length = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.divide(nrm, length, out=np.zeros_like(nrm), where=length > 0)
    rec = np.zeros(len(v), dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["n"], rec["v"] = nrm, v
    with open(filename, "wb") as fh:
        fh.write(b"xlb_amd".ljust(80, b" "))
        fh.write(struct.pack("<I", len(v)))
        fh.write(rec.tobytes())


class VoxelGrid:
    """What ``voxelize_stl`` returns: ``matrix`` (bool occupancy), ``origin`` (coordinates of voxel (0, 0, 0)'s centre), ``pitch``."""

    def __init__(self, matrix, origin, pitch):
        self.matrix, self.origin, self.pitch = matrix, origin, pitch

    @property
    def shape(self):
        return self.matrix.shape

    @property
    def points(self):
        """Centres of the filled voxels, (n, 3), in the mesh's coordinates."""
        return np.argwhere(self.matrix) * self.pitch + self.origin


def voxelize_stl(stl_filename, length_lbm_unit=None, transformation_matrix=None, pitch=None):
    """(VoxelGrid, pitch) of an STL mesh: the voxel size is ``pitch``, or the mesh's largest extent / ``length_lbm_unit``.  A voxel is
    filled when its centre is inside the (closed) surface — the generalised winding number of this backend's WINDING voxeliser — or
    when it overlaps the surface (AABB test)."""
    if length_lbm_unit is None and pitch is None:
        raise ValueError("Either 'length_lbm_unit' or 'pitch' must be provided!")
    verts = load_stl(stl_filename).astype(np.float64)
    extent = (verts.max(axis=0) - verts.min(axis=0)).max()
    if transformation_matrix is not None:
        m = np.asarray(transformation_matrix, dtype=np.float64)
        verts = verts @ m[:3, :3].T + (m[:3, 3] if m.shape == (4, 4) else 0.0)
    if pitch is None:
        pitch = extent / length_lbm_unit
    lo = verts.min(axis=0)
    pad = 2
    local = (verts - lo) / pitch + pad  # lattice units, `pad` empty voxels around the body
    shape = tuple(int(np.ceil(e)) + 2 * pad + 1 for e in local.max(axis=0) - pad)
    from ..grid import grid_factory
    from ..helper import create_nse_fields
    from ..operator.boundary_condition import HalfwayBounceBackBC
    from ..operator.boundary_masker import BC_SOLID, MeshVoxelizationMethod, mesh_masker_for
    from ..velocity_set import D3Q19
    from ..default_config import DefaultConfig

    vs = D3Q19(DefaultConfig.default_precision_policy, DefaultConfig.default_backend)
    grid = grid_factory(shape, velocity_set=vs)
    _, _, f_1, missing_mask, bc_mask = create_nse_fields(grid=grid, velocity_set=vs)
    solid = np.zeros(shape, dtype=bool)
    for method in ("WINDING", "AABB"):
        bc = HalfwayBounceBackBC(velocity_set=vs, mesh_vertices=local.astype(np.float32), voxelization_method=MeshVoxelizationMethod(method))
        bc_mask.fill(0)
        missing_mask.fill(0)
        masker = mesh_masker_for(bc.voxelization_method, vs, DefaultConfig.default_precision_policy, DefaultConfig.default_backend)
        _, bm, _ = masker(bc, f_1, bc_mask, missing_mask)
        solid |= bm.numpy()[0] == BC_SOLID
    return VoxelGrid(solid, lo - pad * pitch, pitch), pitch


# ---- units ---------------------------------------------------------------------------------------------------------------------

class UnitConvertor:
    """Lattice <-> physical units from one velocity pair and the voxel size (utils.py:450-537): dt = dx u_lbm / u_phys.  Every quantity
    has ONE scale — its value of one lattice unit in physical units — and the pair of methods ``<quantity>_to_lbm`` /
    ``<quantity>_to_physical`` divides / multiplies by it: length dx, time dt, velocity dx / dt, viscosity dx^2 / dt, density the
    reference density.  Pressure is the exception: ``pressure_to_lbm`` returns the perturbation p' / (rho_ref u_ref^2) about the reference
    pressure, ``pressure_to_physical`` takes a lattice pressure rho c_s^2 (1/3 at the reference state)."""

    _SCALES = {
        "length": lambda c: c.reference_length,
        "time": lambda c: c.reference_time,
        "density": lambda c: c.reference_density,
        "velocity": lambda c: c.reference_velocity,
        "viscosity": lambda c: c.reference_length**2 / c.reference_time,
    }

    def __init__(self, velocity_lbm_unit, velocity_physical_unit, voxel_size_physical_unit, density_physical_unit=1.2041,
                 pressure_physical_unit=1.101325e5):
        self.voxel_size = voxel_size_physical_unit
        self.velocity_lbm_unit = velocity_lbm_unit
        self.velocity_phys_unit = velocity_physical_unit
        self.reference_density = density_physical_unit
        self.reference_pressure = pressure_physical_unit
        self.reference_length = voxel_size_physical_unit
        self.time_step_physical = self.reference_time = voxel_size_physical_unit * velocity_lbm_unit / velocity_physical_unit
        self.reference_velocity = self.reference_length / self.reference_time

    def pressure_to_lbm(self, pressure_phys):
        return (pressure_phys - self.reference_pressure) / (self.reference_density * self.reference_velocity**2)

    def pressure_to_physical(self, pressure_lbm):
        return self.reference_pressure + (pressure_lbm - 1.0 / 3.0) * self.reference_density * self.reference_velocity**2


def _add_conversions(cls):
    for quantity, scale in cls._SCALES.items():
        setattr(cls, f"{quantity}_to_lbm", lambda self, value, _s=scale: value / _s(self))
        setattr(cls, f"{quantity}_to_physical", lambda self, value, _s=scale: value * _s(self))


_add_conversions(UnitConvertor)
