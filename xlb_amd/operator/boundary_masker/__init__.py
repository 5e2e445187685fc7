"""Boundary maskers of the HIP backend: index lists, and AABB voxelisation of a triangle mesh."""

from .indices_boundary_masker import IndicesBoundaryMasker as IndicesBoundaryMasker
from .mesh_boundary_masker import (
    MeshMaskerAABB as MeshMaskerAABB,
    MeshMaskerRay as MeshMaskerRay,
    MeshVoxelizationMethod as MeshVoxelizationMethod,
    BC_SOLID as BC_SOLID,
)
