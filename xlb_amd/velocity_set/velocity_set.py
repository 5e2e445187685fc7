"""Lattice velocity sets (host tables).

Same public attributes as the reference's `VelocitySet`
(xlb/velocity_set/velocity_set.py:63-83,139-253): ``d, q, c (d,q), w (q), opp_indices, cc,
c_float, qi, cs, cs2, inv_cs2, main_indices, right_indices, left_indices, center_index``.
All tables are NumPy arrays; the kernels carry their own compile-time copy of ``c``/``w``
(csrc/lattice.hpp) and ``tests/test_capi_symbols.py`` / ``tests/test_gpu_*`` check that both
derivations agree through ``xlbhip_lattice_info``.
"""

import math

import numpy as np

from ..compute_backend import ComputeBackend
from ..precision_policy import PrecisionPolicy


class VelocitySet:
    hip_id = None  # lattice id of the C ABI (include/xlbhip.h)

    def __init__(self, d, q, c, w, precision_policy, compute_backend):
        if not isinstance(compute_backend, ComputeBackend):
            raise ValueError(f"Unsupported compute backend: {compute_backend}")
        if not isinstance(precision_policy, PrecisionPolicy):
            raise ValueError(f"Unsupported precision policy: {precision_policy}")
        self.d, self.q = int(d), int(q)
        self.precision_policy = precision_policy
        self.compute_backend = compute_backend

        self._c = np.array(c)
        self._w = np.array(w, dtype=np.float64)
        assert self._c.shape == (self.d, self.q) and self._w.shape == (self.q,)
        self._opp_indices = self._opposites()
        self._cc = self._moment_products()
        self._c_float = self._c.astype(np.float64)
        self._qi = self._q_tensor()

        dtype = precision_policy.compute_precision.np_dtype
        self.c = self._c.astype(np.int32)
        self.w = self._w.astype(dtype)
        self.opp_indices = self._opp_indices.astype(np.int32)
        self.cc = self._cc.astype(dtype)
        self.c_float = self._c_float.astype(dtype)
        self.qi = self._qi.astype(dtype)
        self.cs = dtype(math.sqrt(3) / 3.0)
        self.cs2 = dtype(1.0 / 3.0)
        self.inv_cs2 = dtype(3.0)

        norm1 = np.abs(self._c).sum(axis=0)
        self.main_indices = np.nonzero(norm1 == 1)[0]
        self.right_indices = np.nonzero(self._c[0] == 1)[0]
        self.left_indices = np.nonzero(self._c[0] == -1)[0]
        self.center_index = int(np.nonzero(norm1 == 0)[0][0])

    def _opposites(self):
        cols = [tuple(v) for v in self._c.T.tolist()]
        return np.array([cols.index(tuple(-x for x in v)) for v in cols])

    def _moment_products(self):
        pairs = [(a, b) for a in range(self.d) for b in range(a, self.d)]
        return np.stack([self._c[a] * self._c[b] for a, b in pairs], axis=1).astype(np.float64)

    def _q_tensor(self):
        qi = self._cc.copy()
        k = 0
        for a in range(self.d):
            for b in range(a, self.d):
                if a == b:
                    qi[:, k] -= 1.0 / 3.0
                else:
                    qi[:, k] *= 2.0
                k += 1
        return qi

    def __repr__(self):
        return f"D{self.d}Q{self.q}"

    __str__ = __repr__
