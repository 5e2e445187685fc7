"""HIP grid: allocates device fields (reference template: xlb/grid/warp_grid.py:17-35).

``create_field(cardinality, dtype: Precision = None, fill_value=None)`` returns a
:class:`xlb_amd._lib.Field` whose host view is ``(cardinality, *shape)`` C-order.

Slab decomposition (one process per GPU): when the process context has ``n_ranks > 1`` (see
``xlb_amd.distribute``) or ``backend_config={"halo": True}`` is given, ``shape`` is the GLOBAL
domain, this rank owns the x-planes ``[x_offset, x_offset + local_shape[0])`` and fields carry
ghost planes: TWO per side by default (the two-steps-per-pass kernel recomputes f(t+1) on the inner
one), ``backend_config={"halo": 1}`` for the minimum a single step needs.
"""

from .. import _lib
from ..compute_backend import ComputeBackend
from ..default_config import DefaultConfig, get_context
from ..precision_policy import Precision
from .grid import Grid


def slab_bounds(nx, rank, n_ranks):
    """Contiguous x-plane ranges, remainder spread over the first ranks."""
    base, rem = divmod(int(nx), int(n_ranks))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class HipGrid(Grid):
    def __init__(self, shape, backend_config=None):
        self.backend_config = dict(backend_config or {})
        super().__init__(shape, ComputeBackend.HIP)

    def _initialize_backend(self):
        self.context = get_context()
        rank = self.backend_config.get("rank", self.context.rank)
        n_ranks = self.backend_config.get("n_ranks", self.context.n_ranks)
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        want = self.backend_config.get("halo", None)
        if want is None or want is False:
            self.halo = 2 if self.n_ranks > 1 else 0
        else:
            self.halo = 2 if want is True else int(want)
        if self.halo not in (0, 1, 2) or (self.n_ranks > 1 and self.halo == 0):
            raise ValueError(f"halo={want!r}: slab-decomposed fields need 1 or 2 ghost planes")
        if self.halo and self.dim != 3:
            raise ValueError("slab decomposition needs a 3-D grid")
        if self.n_ranks > 1:
            if self.shape[0] < self.n_ranks:
                raise ValueError(f"cannot split {self.shape[0]} x-planes over {self.n_ranks} ranks")
            self.x_offset, nxl = slab_bounds(self.shape[0], self.rank, self.n_ranks)
            self.local_shape = (nxl,) + self.shape[1:]
        else:
            self.x_offset, self.local_shape = 0, self.shape

    def create_field(self, cardinality, dtype=None, fill_value=None):
        dtype = dtype or DefaultConfig.default_precision_policy.store_precision
        if not isinstance(dtype, Precision):
            raise ValueError(f"dtype must be a Precision, got {dtype!r}")
        return _lib.Field(self.context, cardinality, self.local_shape, dtype.hip_dtype, halo=self.halo, fill_value=fill_value)

    def create_missing_mask(self, cardinality):
        """The missing_mask field: (q, ...) uint8 0/1 on the host, one bit-set per cell on the device."""
        return _lib.Field(self.context, cardinality, self.local_shape, _lib.MISSING, halo=self.halo)
