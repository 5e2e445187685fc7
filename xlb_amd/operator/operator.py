"""Operator base: backend registry + dispatch + dtype policy.

This is the reference's drop-in boundary (xlb/operator/operator.py:35-188), re-stated for a
single registered backend.  Semantics kept:
  * ``@Operator.register_backend(ComputeBackend.X)`` files a method under the key
    ``(class name, backend, signature)`` (operator.py:74-87);
  * ``op(*args, callback=None, **kwargs)`` tries every candidate of the instance's class and
    backend, binds the arguments against its signature and returns the first that succeeds;
    if none is registered it raises ``NotImplementedError``; if all raise, a plain
    ``Exception`` carrying the last error and traceback (operator.py:89-133);
  * ``compute_dtype`` / ``store_dtype`` come from the precision policy (operator.py:166-188).
"""

import inspect
import traceback

from ..compute_backend import ComputeBackend
from ..default_config import DefaultConfig, get_context


class Operator:
    _backends = {}

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None):
        self.velocity_set = velocity_set or DefaultConfig.velocity_set
        self.precision_policy = precision_policy or DefaultConfig.default_precision_policy
        self.compute_backend = compute_backend or DefaultConfig.default_backend
        if not isinstance(self.compute_backend, ComputeBackend):
            raise ValueError(f"Compute_backend {compute_backend} is not supported")
        if self.compute_backend is ComputeBackend.HIP:
            self._construct_hip()

    def _construct_hip(self):
        """Hook run at construction for the HIP backend (the reference's _construct_warp slot,
        operator.py:62-66).  The kernels are precompiled, so the default is a no-op."""

    @classmethod
    def register_backend(cls, backend_name):
        def decorator(func):
            owner = func.__qualname__.split(".")[0]
            cls._backends[(owner, backend_name, str(inspect.signature(func)))] = func
            return func

        return decorator

    def __call__(self, *args, callback=None, **kwargs):
        name = self.__class__.__name__
        candidates = [(k, m) for k, m in self._backends.items() if k[0] == name and k[1] == self.compute_backend]
        if not candidates:
            supported = [k for k in self._backends if k[0] == name]
            raise NotImplementedError(
                f"No implementation found for operator {name} with backend {self.compute_backend}. Available implementations: {supported}"
            )
        key = error = tb = None
        for key, method in candidates:
            try:
                inspect.signature(method).bind(self, *args, **kwargs).apply_defaults()
                result = method(self, *args, **kwargs)
                if callback and callable(callback):
                    callback(result if result is not None else (args, kwargs))
                return result
            except Exception as e:  # the reference swallows everything and tries the next candidate
                error, tb = e, traceback.format_exc()
        raise Exception(f"Error captured for backend with key {key} for operator {name}: {error}\n {tb}")

    @property
    def supported_compute_backend(self):
        return list(self._backends.keys())

    def __repr__(self):
        return f"{self.__class__.__name__}()"

    @property
    def compute_dtype(self):
        return self.precision_policy.compute_precision.np_dtype

    @property
    def store_dtype(self):
        return self.precision_policy.store_precision.np_dtype

    # C-ABI codes
    @property
    def _compute_code(self):
        return self.precision_policy.compute_precision.hip_dtype

    @property
    def _store_code(self):
        return self.precision_policy.store_precision.hip_dtype

    @property
    def _ctx(self):
        return get_context()

    def get_precision_policy(self):
        return self.precision_policy
