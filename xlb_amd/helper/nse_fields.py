"""The four fields of a Navier-Stokes run (reference xlb/helper/nse_fields.py:16-55):
``f_0, f_1`` (q, ...) in the store dtype, ``bc_mask`` (1, ...) uint8 and ``missing_mask``
(q, ...) — uint8 0/1 on the host as on the reference's kernel backends, bit-packed on the
device."""

from ..default_config import DefaultConfig
from ..grid import grid_factory
from ..precision_policy import Precision


def _or_default(value, name):
    return getattr(DefaultConfig, name) if value is None else value


def create_nse_fields(grid_shape=None, grid=None, velocity_set=None, compute_backend=None, precision_policy=None):
    """-> (grid, f_0, f_1, missing_mask, bc_mask); the grid is built from ``grid_shape`` unless one is passed in."""
    lattice = _or_default(velocity_set, "velocity_set")
    backend = _or_default(compute_backend, "default_backend")
    store = _or_default(precision_policy, "default_precision_policy").store_precision
    if grid is None and grid_shape is None:
        raise ValueError("grid_shape must be provided when grid is None")
    grid = grid or grid_factory(grid_shape, compute_backend=backend, velocity_set=lattice)
    populations = [grid.create_field(lattice.q, store) for _ in range(2)]  # f_0, f_1 (the A/B pair)
    masks = {"missing": grid.create_missing_mask(lattice.q), "bc": grid.create_field(1, Precision.UINT8)}
    return (grid, *populations, masks["missing"], masks["bc"])
