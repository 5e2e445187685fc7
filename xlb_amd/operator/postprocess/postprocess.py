"""Vorticity and Q-criterion from a velocity field (reference xlb/operator/postprocess/vorticity.py and
q_criterion.py; the reference implements them for its kernel backend only — the JAX methods raise
NotImplementedError — with the out-of-place call style ``op(u, bc_mask, out_a, out_b) -> (out_a, out_b)``).

3-D fields.  Cells one layer inside the box whose six face neighbours carry no boundary id are written; the
outputs keep their previous contents everywhere else (create them zero-filled, as the reference's callers do)."""

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class Vorticity(Operator):
    """``Vorticity()(u, bc_mask, vorticity, vorticity_magnitude) -> (vorticity, vorticity_magnitude)``: curl of u by
    central differences and its Euclidean norm (vorticity.py:63-84)."""

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, u, bc_mask, vorticity, vorticity_magnitude):
        _lib.check(_lib.load().xlbhip_vorticity(self._ctx.handle, u.handle, bc_mask.handle, vorticity.handle, vorticity_magnitude.handle))
        return vorticity, vorticity_magnitude


class QCriterion(Operator):
    """``QCriterion()(u, bc_mask, norm_mu, q) -> (norm_mu, q)``: vorticity magnitude and the second invariant of the
    velocity gradient, Q = (|Omega|^2 - |S|^2) / 2 (q_criterion.py:66-131)."""

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, u, bc_mask, norm_mu, q):
        _lib.check(_lib.load().xlbhip_q_criterion(self._ctx.handle, u.handle, bc_mask.handle, norm_mu.handle, q.handle))
        return norm_mu, q


class GridToPoint(Operator):
    """``GridToPoint()(grid, points, point_values) -> point_values``: trilinear interpolation of component 0 of a 3-D field
    at arbitrary points (grid_to_point.py:28-104).  ``points`` is an ``(n, 3)`` float32 array in cell units,
    ``point_values`` an ``(n,)`` array of the field's dtype that receives the result (host arrays: probes are few)."""

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, grid, points, point_values):
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must have shape (n, 3)")
        if not (isinstance(point_values, np.ndarray) and point_values.flags.c_contiguous and point_values.shape == (pts.shape[0],) and point_values.dtype == grid.dtype):
            raise ValueError("point_values must be a contiguous (n,) array of the field's dtype")
        _lib.check(_lib.load().xlbhip_grid_to_point(self._ctx.handle, grid.handle, int(pts.shape[0]), pts.ctypes.data, point_values.ctypes.data))
        return point_values

