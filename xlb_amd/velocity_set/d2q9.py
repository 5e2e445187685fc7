import numpy as np

from .velocity_set import VelocitySet


class D2Q9(VelocitySet):
    """Direction order of the reference (xlb/velocity_set/d2q9.py:18-21): rest, the four axis
    directions and four diagonals in its hand-listed sequence."""

    hip_id = 0

    def __init__(self, precision_policy, compute_backend):
        c = np.array([[0, 0, 0, 1, -1, 1, -1, 1, -1], [0, 1, -1, 0, 1, -1, 0, 1, -1]])
        n1 = np.abs(c).sum(axis=0)
        w = np.choose(n1, [4 / 9, 1 / 9, 1 / 36])
        super().__init__(2, 9, c, w, precision_policy, compute_backend)
