"""Grids of the HIP backend."""

from .grid import grid_factory as grid_factory, Grid as Grid
from .hip_grid import HipGrid as HipGrid
