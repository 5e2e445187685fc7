"""Mesh-based boundary masker of the HIP backend: AABB voxelisation (reference
xlb/operator/boundary_masker/{mesh_boundary_masker,aabb,mesh_voxelization_method}.py; the reference has the mesh maskers for
its kernel backends only, with the call ``masker(bc, f_1, bc_mask, missing_mask) -> (f_1, bc_mask, missing_mask)``).

``bc.mesh_vertices``: triangle soup, ``(3 n_triangles, 3)`` in lattice units (voxel ``i`` spans ``[i, i+1]``), inside the
domain.  Voxels the surface passes through become ``BC_SOLID`` (255), fluid voxels next to one get the BC's id and the
missing bits of the directions pulled out of the solid (AABB); or the voxels whose lattice links cross the surface get them
(RAY).  The WINDING / AABB_CLOSE methods are not built."""

from dataclasses import dataclass, field

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator

BC_SOLID = 255
METHODS = {"AABB": 1, "RAY": 2, "AABB_CLOSE": 3, "WINDING": 4}


@dataclass
class VoxelizationMethod:
    id: int
    name: str
    options: dict = field(default_factory=dict)


def MeshVoxelizationMethod(name, **options):
    assert name in METHODS, f"Unsupported voxelization method: {name}"
    return VoxelizationMethod(METHODS[name], name, options)


class _MeshMasker(Operator):
    _entry = None  # name of the C entry point

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None):
        super().__init__(velocity_set, precision_policy, compute_backend)
        assert self.velocity_set.d == 3, "MeshBoundaryMasker is only implemented for 3D velocity sets!"

    def _mask(self, bc, distances, bc_mask, missing_mask):
        assert bc.mesh_vertices is not None, f'Please provide the mesh vertices for {bc.__class__.__name__} BC using keyword "mesh_vertices"!'
        assert bc.indices is None, f"Please use IndicesBoundaryMasker operator if {bc.__class__.__name__} is imposed on known indices of the grid!"
        if getattr(bc, "needs_mesh_distance", False):
            raise NotImplementedError("mesh distances (HybridBC) are out of scope of the HIP backend")
        verts = np.ascontiguousarray(bc.mesh_vertices, dtype=np.float32)
        grid_shape = bc_mask.grid_shape
        lo, hi = verts.min(axis=0), verts.max(axis=0)
        if np.any(lo < 0) or np.any(hi >= np.array(grid_shape)):
            raise ValueError(
                f"Mesh extents ({lo}, {hi}) exceed domain dimensions {grid_shape}. The mesh must be fully contained within the domain."
            )
        bc.__dict__["mesh_vertices"] = None  # consumed, like the reference (mesh_boundary_masker.py:204)
        entry = getattr(_lib.load(), self._entry)
        _lib.check(entry(self._ctx.handle, self.velocity_set.hip_id, int(bc.id), int(verts.shape[0] // 3), verts.ctypes.data, bc_mask.handle, missing_mask.handle))
        return distances, bc_mask, missing_mask


class MeshMaskerAABB(_MeshMasker):
    """Surface voxels (triangle / box overlap) become BC_SOLID, their fluid neighbours the boundary voxels (aabb.py:38-100)."""

    _entry = "xlbhip_mesh_mask_aabb"

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)


class MeshMaskerRay(_MeshMasker):
    """Voxels whose lattice links cross the surface are the boundary voxels; nothing is marked solid (ray.py:38-76)."""

    _entry = "xlbhip_mesh_mask_ray"

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)
