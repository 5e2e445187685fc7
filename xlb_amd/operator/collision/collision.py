"""Collision operators BGK and KBC as stand-alone whole-field operators
(reference xlb/operator/collision/bgk.py:27-32,:78-91 and kbc.py:40-79)."""

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class Collision(Operator):
    hip_collision_id = None

    def _launch(self, f, feq, fout, omega):
        _lib.check(
            _lib.load().xlbhip_collide(
                self._ctx.handle, self.velocity_set.hip_id, self.hip_collision_id, self._compute_code, f.handle, feq.handle, fout.handle, float(omega)
            )
        )
        return fout


class BGK(Collision):
    """fout = f - omega (f - feq)"""

    hip_collision_id = _lib.BGK

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, feq, fout, omega):
        return self._launch(f, feq, fout, omega)


class KBC(Collision):
    """Entropic multi-relaxation (Karlin-Boesch-Chikatamarla) collision; D2Q9 and D3Q27 only."""

    hip_collision_id = _lib.KBC

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None):
        self.epsilon = 1e-32
        super().__init__(velocity_set, precision_policy, compute_backend)

    def _construct_hip(self):
        if self.velocity_set.hip_id not in (_lib.D2Q9, _lib.D3Q27):
            raise NotImplementedError(f"Velocity set not supported: {type(self.velocity_set)}")

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, feq, fout, omega):
        return self._launch(f, feq, fout, omega)


class SmagorinskyLESBGK(Collision):
    """BGK with the Smagorinsky sub-grid model: the relaxation time is raised by the local strain
    rate estimated from the non-equilibrium stress (reference
    xlb/operator/collision/smagorinsky_les_bgk.py:17-60)."""

    hip_collision_id = _lib.SMAGORINSKY_LES_BGK

    def __init__(self, velocity_set=None, precision_policy=None, compute_backend=None, smagorinsky_coef: float = 0.17):
        self.smagorinsky_coef = smagorinsky_coef
        super().__init__(velocity_set, precision_policy, compute_backend)

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f, feq, fout, omega):
        self._ctx.set_option("smagorinsky_coef_e6", round(self.smagorinsky_coef * 1e6))
        return self._launch(f, feq, fout, omega)
