"""Precision / PrecisionPolicy, the `<compute><store>` pairs of the reference
(xlb/precision_policy.py:13-110), with accessors for this backend: a NumPy dtype for host
views and the element-type code of the C ABI (include/xlbhip.h)."""

from enum import Enum

import numpy as np

# (numpy dtype, xlbhip dtype code)
_TABLE = {
    "FP64": (np.float64, 0),
    "FP32": (np.float32, 1),
    "FP16": (np.float16, 2),
    "UINT8": (np.uint8, 3),
    "BOOL": (np.bool_, 4),
}


class Precision(Enum):
    FP64 = 1
    FP32 = 2
    FP16 = 3
    UINT8 = 4
    BOOL = 5

    @property
    def np_dtype(self):
        return _TABLE[self.name][0]

    @property
    def hip_dtype(self):
        return _TABLE[self.name][1]


class PrecisionPolicy(Enum):
    FP64FP64 = 1
    FP64FP32 = 2
    FP64FP16 = 3
    FP32FP32 = 4
    FP32FP16 = 5

    @property
    def compute_precision(self):
        return Precision[self.name[:4]]

    @property
    def store_precision(self):
        return Precision[self.name[4:]]

    def cast_to_compute_np(self, array):
        return np.asarray(array, dtype=self.compute_precision.np_dtype)

    def cast_to_store_np(self, array):
        return np.asarray(array, dtype=self.store_precision.np_dtype)
