import itertools

import numpy as np

from .velocity_set import VelocitySet


class D3Q19(VelocitySet):
    """itertools.product([0, -1, 1], repeat=3) filtered to |c|_1 <= 2
    (reference xlb/velocity_set/d3q19.py:19-27)."""

    hip_id = 1

    def __init__(self, precision_policy, compute_backend):
        full = np.array(list(itertools.product([0, -1, 1], repeat=3)))
        c = full[np.abs(full).sum(axis=1) <= 2].T
        n1 = np.abs(c).sum(axis=0)
        w = np.choose(n1, [1 / 3, 1 / 18, 1 / 36])
        super().__init__(3, 19, c, w, precision_policy, compute_backend)
