"""Moments of the populations as stand-alone operators: rho = sum f, u = (sum c f) / rho,
Pi = sum cc f (reference xlb/operator/macroscopic/{zero_moment,first_moment,macroscopic,
second_moment}.py; kernel-backend call style macroscopic.py:57-64)."""

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


def _macro(op, f, rho, u):
    _lib.check(
        _lib.load().xlbhip_macroscopic(
            op._ctx.handle, op.velocity_set.hip_id, op._compute_code, f.handle, rho.handle if rho is not None else None, u.handle if u is not None else None
        )
    )


class ZeroMoment(Operator):
    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, rho):
        _macro(self, f, rho, None)
        return rho


class FirstMoment(Operator):
    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, u):
        _macro(self, f, None, u)
        return u


class Macroscopic(Operator):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.zero_moment = ZeroMoment(self.velocity_set, self.precision_policy, self.compute_backend)
        self.first_moment = FirstMoment(self.velocity_set, self.precision_policy, self.compute_backend)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, rho, u):
        _macro(self, f, rho, u)
        return rho, u


class SecondMoment(Operator):
    """Pi components in the order (xx, xy, xz, yy, yz, zz) / (xx, xy, yy)."""

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, pi):
        _lib.check(_lib.load().xlbhip_second_moment(self._ctx.handle, self.velocity_set.hip_id, self._compute_code, f.handle, pi.handle))
        return pi
