"""Builds bc_mask / missing_mask from explicit index lists.

Bit-exact target: the reference's JAX masker (xlb/operator/boundary_masker/
indices_boundary_masker.py:73-143): ids are scattered in list order (later BCs overwrite
earlier ones); if ANY index of a BC is strictly interior, the BC's indices are treated as solid
cells (all their populations marked missing) and bc_mask is tagged at ``bc.pad_indices()``;
finally the (outside | solid | previous) marks are pull-streamed, so that
``missing_mask[l, x]`` is set iff ``x - c_l`` is outside the domain or solid — for EVERY cell.

Host side (this file): the interior test and ``pad_indices`` (index bookkeeping, O(N_idx)).
Device side (xlbhip_build_masks): O(N_idx) scatters + one O(N q) neighbour test, instead of the
reference Warp kernels' O(N * N_idx) scan (indices_boundary_masker.py:160-163).
"""

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class IndicesBoundaryMasker(Operator):
    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, grid=None):
        super().__init__(velocity_set, precision_policy, compute_backend)
        self.grid = grid

    def are_indices_in_interior(self, indices, shape):
        d = self.velocity_set.d
        sh = np.array(shape)
        return np.all((indices[:d] > 0) & (indices[:d] < sh[:d, np.newaxis] - 1), axis=0)

    @staticmethod
    def _as3(idx):
        idx = np.asarray(idx, dtype=np.int64)
        if idx.shape[0] == 2:  # (x, y) -> internal (0, x, y)
            idx = np.concatenate([np.zeros((1, idx.shape[1]), np.int64), idx], axis=0)
        return np.ascontiguousarray(idx, dtype=np.int32)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, bclist, bc_mask, missing_mask, start_index=None):
        grid = self.grid
        if grid is None:
            raise ValueError("IndicesBoundaryMasker needs the grid on the HIP backend")
        gshape = grid.shape
        x_offset = grid.x_offset if start_index is None else int(start_index[0])
        ids, tags, solids = [], [], []
        for bc in bclist:
            assert bc.indices is not None, f"Please specify indices associated with the {bc.__class__.__name__} BC!"
            idx = np.array(bc.indices)
            if idx.ndim != 2 or idx.shape[0] != self.velocity_set.d:
                raise ValueError(f"{bc.__class__.__name__}.indices must be {self.velocity_set.d} lists of equal length")
            if np.any(self.are_indices_in_interior(idx, gshape)):
                solids.append(self._as3(idx))
                tags.append(self._as3(bc.pad_indices()))
            else:
                solids.append(None)
                tags.append(self._as3(idx))
            ids.append(bc.id)
            # the reference drops the (large) index lists once consumed (indices_boundary_masker.py:131)
            bc.__dict__.pop("indices", None)
        g3 = (1,) + tuple(gshape) if len(gshape) == 2 else tuple(gshape)
        _lib.build_masks(self._ctx, self.velocity_set.hip_id, ids, tags, solids, g3, x_offset, bc_mask, missing_mask)
        return bc_mask, missing_mask
