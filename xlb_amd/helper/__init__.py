"""Set-up helpers of the HIP backend: field allocation, equilibrium initialisation, overlap check."""

from .nse_fields import create_nse_fields as create_nse_fields
from .initializers import initialize_eq as initialize_eq
from .check_boundary_overlaps import check_bc_overlaps as check_bc_overlaps
