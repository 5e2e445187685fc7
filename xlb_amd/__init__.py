"""xlb_amd — MI355X-native compute backend for the XLB lattice-Boltzmann operator API.

Same import surface as the slice of ``xlb`` the per-timestep hot path needs
(reference xlb/__init__.py:9-38): enums, ``init``, velocity sets, grid factory, operators,
boundary conditions, the stepper and helpers.  Everything numerical runs in libxlbhip.so
(hand-written HIP for gfx950) through the C ABI declared in include/xlbhip.h.
"""

__version__ = "0.1.0"

from .compute_backend import ComputeBackend as ComputeBackend
from .precision_policy import PrecisionPolicy as PrecisionPolicy, Precision as Precision
from .default_config import init as init, DefaultConfig as DefaultConfig

from . import velocity_set as velocity_set
from . import grid as grid
from . import operator as operator
from .operator import equilibrium as _equilibrium  # noqa: F401
from .operator import collision as _collision  # noqa: F401
from .operator import stream as _stream  # noqa: F401
from .operator import macroscopic as _macroscopic  # noqa: F401
from .operator import boundary_condition as _boundary_condition  # noqa: F401
from .operator import boundary_masker as _boundary_masker  # noqa: F401
from .operator import stepper as _stepper  # noqa: F401
from . import helper as helper
from . import distribute as distribute
from . import utils as utils
