"""Process-global defaults, set once by :func:`init` (reference xlb/default_config.py:16-100).

For the HIP backend ``init`` also opens the device context (the analogue of the reference's
``wp.init()`` + device pin, default_config.py:38-57).  The device is chosen by, in order:
the ``XLB_HIP_DEVICE`` environment variable, ``LOCAL_RANK`` (one process per GPU), 0.
"""

import os

from .compute_backend import ComputeBackend


class DefaultConfig:
    default_precision_policy = None
    velocity_set = None
    default_backend = None
    # HIP backend state
    context = None


def _pick_device():
    for var in ("XLB_HIP_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var, "").strip()
        if v:
            return int(v.split(":")[-1])
    return 0


def init(velocity_set, default_backend, default_precision_policy):
    DefaultConfig.velocity_set = velocity_set
    DefaultConfig.default_backend = default_backend
    DefaultConfig.default_precision_policy = default_precision_policy
    if default_backend is ComputeBackend.HIP:
        get_context()
    elif isinstance(default_backend, ComputeBackend):
        raise ValueError(f"Compute backend {default_backend} is not available in xlb_amd; use ComputeBackend.HIP")
    else:
        raise ValueError(f"Unsupported compute backend: {default_backend}")


def get_context():
    """The process's device context (created on first use)."""
    if DefaultConfig.context is None:
        from ._lib import Context

        DefaultConfig.context = Context(_pick_device())
    return DefaultConfig.context


def default_backend():
    return DefaultConfig.default_backend
