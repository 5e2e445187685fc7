"""Mesh-based boundary masker of the HIP backend: AABB voxelisation (reference
xlb/operator/boundary_masker/{mesh_boundary_masker,aabb,mesh_voxelization_method}.py; the reference has the mesh maskers for
its kernel backends only, with the call ``masker(bc, f_1, bc_mask, missing_mask) -> (f_1, bc_mask, missing_mask)``).

``bc.mesh_vertices``: triangle soup, ``(3 n_triangles, 3)`` in lattice units (voxel ``i`` spans ``[i, i+1]``), inside the
domain.  Voxels the surface passes through become ``BC_SOLID`` (255), fluid voxels next to one get the BC's id and the
missing bits of the directions pulled out of the solid (AABB); or the voxels whose lattice links cross the surface get them
(RAY); WINDING marks the voxels whose centre lies inside the mesh (generalized winding number) and tags their fluid
neighbours; AABB_CLOSE closes the AABB shell morphologically (dilate, erode) first (winding.py, aabb_close.py).

Wall distances: when ``bc.needs_mesh_distance`` (HybridBC(use_mesh_distance=True)) the masker also computes the fractional
distance to the surface along every cut link.  The reference writes them into the `distances` argument — its stepper passes
``f_1`` — and recovers them from there every step; here they are gathered into ``bc._distance_table`` (cells, weights),
which the stepper hands to the native stepper's sorted table: the `distances` argument is returned untouched."""

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
    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None):
        super().__init__(velocity_set, precision_policy, compute_backend)
        assert self.velocity_set.d == 3, "MeshBoundaryMasker is only implemented for 3D velocity sets!"

    _method = None  # _lib.MESH_*
    close_voxels = 0

    def _mask(self, bc, distances, bc_mask, missing_mask):
        assert bc.mesh_vertices is not None, f'Please provide the mesh vertices for {bc.__class__.__name__} BC using keyword "mesh_vertices"!'
        assert bc.indices is None, f"Please use IndicesBoundaryMasker operator if {bc.__class__.__name__} is imposed on known indices of the grid!"
        want_dist = bool(getattr(bc, "needs_mesh_distance", False))
        if want_dist and self._method == _lib.MESH_AABB:
            raise NotImplementedError("MeshMaskerAABB has no wall distances; use RAY, WINDING or AABB_CLOSE with use_mesh_distance")
        verts = np.ascontiguousarray(bc.mesh_vertices, dtype=np.float32)
        grid_shape = bc_mask.grid_shape
        lo, hi = verts.min(axis=0), verts.max(axis=0)
        if np.any(lo < 0) or np.any(hi >= np.array(grid_shape)):
            raise ValueError(
                f"Mesh extents ({lo}, {hi}) exceed domain dimensions {grid_shape}. The mesh must be fully contained within the domain."
            )
        bc.__dict__["mesh_vertices"] = None  # consumed, like the reference (mesh_boundary_masker.py:204)
        dist = None
        if want_dist:
            dist = _lib.Field(self._ctx, self.velocity_set.q, grid_shape, _lib.F32)  # temporary dense (q, ...) weights, zero-filled
        try:
            _lib.check(_lib.load().xlbhip_mesh_mask(self._ctx.handle, self.velocity_set.hip_id, int(self._method), int(bc.id), int(verts.shape[0] // 3),
                                                    verts.ctypes.data, int(self.close_voxels), bc_mask.handle, missing_mask.handle,
                                                    dist.handle if dist is not None else None))
            if want_dist:
                # the boundary voxels of this BC and their q weights (a 1 B / cell download + a gather of the few rows)
                cells = np.flatnonzero(bc_mask.numpy().reshape(-1) == bc.id).astype(np.uint32)
                bc._distance_table = (cells, dist.gather(cells))
        finally:
            if dist is not None:
                dist.free()
        return distances, bc_mask, missing_mask


class MeshMaskerAABB(_MeshMasker):
    """Surface voxels (triangle / box overlap) become BC_SOLID, their fluid neighbours the boundary voxels (aabb.py:38-100)."""

    _method = _lib.MESH_AABB

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)


class MeshMaskerRay(_MeshMasker):
    """Voxels whose lattice links cross the surface are the boundary voxels; nothing is marked solid (ray.py:38-76)."""

    _method = _lib.MESH_RAY

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)


class MeshMaskerWinding(_MeshMasker):
    """Voxels whose centre is inside the mesh (generalized winding number > 0.5) become BC_SOLID; a fluid neighbour reached by
    a lattice link that crosses the surface is a boundary voxel (winding.py:46-103).  Works for non-watertight soups too."""

    _method = _lib.MESH_WINDING

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)


class MeshMaskerAABBClose(_MeshMasker):
    """AABB voxelisation followed by a morphological close (dilate then erode by ``close_voxels`` layers), which seals
    small holes and thin slits of the surface and fills closed shells (aabb_close.py:26-365)."""

    _method = _lib.MESH_AABB_CLOSE

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, close_voxels=None):
        assert close_voxels is not None, (
            "Please provide the number of close voxels using the 'close_voxels' argument! e.g., MeshVoxelizationMethod('AABB_CLOSE', close_voxels=3)"
        )
        self.close_voxels = int(close_voxels)
        super().__init__(velocity_set, precision_policy, compute_backend)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bc, distances, bc_mask, missing_mask):
        return self._mask(bc, distances, bc_mask, missing_mask)


def mesh_masker_for(method, velocity_set=None, precision_policy=None, compute_backend=None):
    """The masker operator of a MeshVoxelizationMethod (None = AABB, the reference's default): nse_stepper.py:165-203."""
    name = "AABB" if method is None else getattr(method, "name", "AABB")
    if name == "AABB":
        return MeshMaskerAABB(velocity_set, precision_policy, compute_backend)
    if name == "RAY":
        return MeshMaskerRay(velocity_set, precision_policy, compute_backend)
    if name == "WINDING":
        return MeshMaskerWinding(velocity_set, precision_policy, compute_backend)
    if name == "AABB_CLOSE":
        return MeshMaskerAABBClose(velocity_set, precision_policy, compute_backend, close_voxels=method.options.get("close_voxels"))
    raise NotImplementedError(f"voxelization method {name}")
