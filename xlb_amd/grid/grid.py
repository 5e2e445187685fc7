"""Grid abstraction + factory (reference xlb/grid/grid.py:19-58, :102-191)."""

import numpy as np

from ..compute_backend import ComputeBackend
from ..default_config import DefaultConfig


def grid_factory(shape, compute_backend=None, velocity_set=None, backend_config=None):
    compute_backend = compute_backend or DefaultConfig.default_backend
    if compute_backend is ComputeBackend.HIP:
        from .hip_grid import HipGrid

        return HipGrid(shape, backend_config=backend_config)
    raise ValueError(f"Compute backend {compute_backend} is not supported")


class Grid:
    def __init__(self, shape, compute_backend):
        self.shape = tuple(int(s) for s in shape)
        self.dim = len(self.shape)
        if self.dim not in (2, 3):
            raise ValueError("grid must be 2-D or 3-D")
        self.compute_backend = compute_backend
        self._initialize_backend()

    def _initialize_backend(self):
        raise NotImplementedError

    def get_compute_backend(self):
        return self.compute_backend

    def bounding_box_indices(self, shape=None, remove_edges=False, as_numpy=False):
        """Index lists of the faces of the box, same contents and ordering as the reference
        (grid.py:135-191): "bottom"/"top" = last axis 0 / n-1, "left"/"right" = x, and in 3-D
        "front"/"back" = y; ``remove_edges`` trims every tangential range by one cell.

        Built from per-face ``np.meshgrid`` ranges instead of a full ``np.indices(shape)``
        array, so a 512^3 domain does not allocate 3.2 GB on the host.  ``as_numpy=True`` (an extension)
        returns ``(dim, n)`` int32 arrays instead of nested Python lists — what large drivers want."""
        shape = tuple(self.shape if shape is None else shape)
        dim = len(shape)
        lo = 1 if remove_edges else 0
        rng = [np.arange(lo, n - lo if remove_edges else n) for n in shape]

        def face(axis, value):
            axes = [np.array([value]) if a == axis else rng[a] for a in range(dim)]
            mesh = np.meshgrid(*axes, indexing="ij")
            if as_numpy:
                return np.stack([m.reshape(-1) for m in mesh]).astype(np.int32)
            return [m.reshape(-1).tolist() for m in mesh]

        last = dim - 1
        box = {
            "bottom": face(last, 0),
            "top": face(last, shape[last] - 1),
            "left": face(0, 0),
            "right": face(0, shape[0] - 1),
        }
        if dim == 3:
            box["front"] = face(1, 0)
            box["back"] = face(1, shape[1] - 1)
        return box
