"""Boundary maskers of the HIP backend: index lists, and the voxelisations of a triangle mesh (AABB, RAY, WINDING, AABB_CLOSE)."""

from .indices_boundary_masker import IndicesBoundaryMasker as IndicesBoundaryMasker
from .mesh_boundary_masker import (
    MeshMaskerAABB as MeshMaskerAABB,
    MeshMaskerRay as MeshMaskerRay,
    MeshMaskerWinding as MeshMaskerWinding,
    MeshMaskerAABBClose as MeshMaskerAABBClose,
    mesh_masker_for as mesh_masker_for,
    MeshVoxelizationMethod as MeshVoxelizationMethod,
    BC_SOLID as BC_SOLID,
)
