"""D2Q9 in the reference's direction order (xlb/velocity_set/d2q9.py:18-21): not the itertools order of the 3-D sets but a
hand-listed one — rest first, then (0, ±1), (±1, 0) and the diagonals interleaved."""

import numpy as np

from .velocity_set import VelocitySet

# direction l = (c_x, c_y); the order is part of the data format (populations are stored in it)
_DIRECTIONS = ((0, 0), (0, 1), (0, -1), (1, 0), (-1, 1), (1, -1), (-1, 0), (1, 1), (-1, -1))
_WEIGHT_BY_SPEED2 = {0: 4.0 / 9.0, 1: 1.0 / 9.0, 2: 1.0 / 36.0}


class D2Q9(VelocitySet):
    hip_id = 0

    def __init__(self, precision_policy, compute_backend):
        c = np.asarray(_DIRECTIONS, dtype=np.int64).T
        w = np.array([_WEIGHT_BY_SPEED2[cx * cx + cy * cy] for cx, cy in _DIRECTIONS])
        super().__init__(2, 9, c, w, precision_policy, compute_backend)
