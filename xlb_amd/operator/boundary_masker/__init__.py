"""Boundary maskers of the HIP backend (index lists only; mesh voxelisation is out of scope)."""

from .indices_boundary_masker import IndicesBoundaryMasker as IndicesBoundaryMasker
