"""Lattice velocity sets available on the HIP backend: D2Q9, D3Q19, D3Q27 (host tables; the kernels carry
their own compile-time copies, see csrc/lattice.hpp)."""

from .velocity_set import VelocitySet as VelocitySet
from .d2q9 import D2Q9 as D2Q9
from .d3q19 import D3Q19 as D3Q19
from .d3q27 import D3Q27 as D3Q27
