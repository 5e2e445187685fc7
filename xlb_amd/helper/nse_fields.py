"""The four fields of a Navier-Stokes run (reference xlb/helper/nse_fields.py:16-55):
``f_0, f_1`` (q, ...) in the store dtype, ``bc_mask`` (1, ...) uint8 and ``missing_mask``
(q, ...) — uint8 0/1 on the host as on the reference's kernel backends, bit-packed on the
device."""

from ..default_config import DefaultConfig
from ..grid import grid_factory
from ..precision_policy import Precision


def create_nse_fields(grid_shape=None, grid=None, velocity_set=None, compute_backend=None, precision_policy=None):
    velocity_set = velocity_set or DefaultConfig.velocity_set
    compute_backend = compute_backend or DefaultConfig.default_backend
    precision_policy = precision_policy or DefaultConfig.default_precision_policy
    if grid is None:
        if grid_shape is None:
            raise ValueError("grid_shape must be provided when grid is None")
        grid = grid_factory(grid_shape, compute_backend=compute_backend, velocity_set=velocity_set)
    f_0 = grid.create_field(cardinality=velocity_set.q, dtype=precision_policy.store_precision)
    f_1 = grid.create_field(cardinality=velocity_set.q, dtype=precision_policy.store_precision)
    bc_mask = grid.create_field(cardinality=1, dtype=Precision.UINT8)
    missing_mask = grid.create_missing_mask(velocity_set.q)
    return grid, f_0, f_1, missing_mask, bc_mask
