"""Compute-backend selector.

Mirrors the role of the reference's `xlb/compute_backend.py:8-18`.  The only backend this
package implements is ``HIP`` (hand-written CDNA4 kernels behind the C ABI of
``include/xlbhip.h``).  The reference's member names are kept so that a driver script that
still says ``ComputeBackend.WARP`` fails with a clear message at ``init`` time instead of an
``AttributeError``.
"""

from enum import Enum


class ComputeBackend(Enum):
    JAX = 1
    WARP = 2
    NEON = 3
    HIP = 4  # MI355X-native backend (this package)

    @property
    def available(self):
        return self is ComputeBackend.HIP
