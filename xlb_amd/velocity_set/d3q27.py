import itertools

import numpy as np

from .velocity_set import VelocitySet


class D3Q27(VelocitySet):
    """Full itertools.product([0, -1, 1], repeat=3) (reference xlb/velocity_set/d3q27.py:19-29)."""

    hip_id = 2

    def __init__(self, precision_policy, compute_backend):
        c = np.array(list(itertools.product([0, -1, 1], repeat=3))).T
        n1 = np.abs(c).sum(axis=0)
        w = np.choose(n1, [8 / 27, 2 / 27, 1 / 54, 1 / 216])
        super().__init__(3, 27, c, w, precision_policy, compute_backend)
