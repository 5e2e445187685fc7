"""Second-order (quadratic) equilibrium, feq = rho w (1 + cu (1 + 0.5 cu) - usqr)
(reference xlb/operator/equilibrium/quadratic_equilibrium.py:23-30; call style :91-103)."""

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class Equilibrium(Operator):
    pass


class QuadraticEquilibrium(Equilibrium):
    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, rho, u, f):
        _lib.check(_lib.load().xlbhip_equilibrium(self._ctx.handle, self.velocity_set.hip_id, self._compute_code, rho.handle, u.handle, f.handle))
        return f

    def host_values(self, rho, u):
        """feq for ONE (rho, u) pair, evaluated on the host in the compute dtype with the same
        operation order as the kernels; used for the constant vectors of EquilibriumBC
        (bc_equilibrium.py:72-74)."""
        vs = self.velocity_set
        T = self.compute_dtype
        rho = T(rho)
        uu = [T(x) for x in u]
        usq = uu[0] * uu[0]
        for d in range(1, vs.d):
            usq = T(usq + uu[d] * uu[d])
        usqr = T(T(1.5) * usq)
        out = np.zeros(vs.q, dtype=T)
        w = vs._w.astype(T)
        for l in range(vs.q):
            dot = T(0)
            for d in range(vs.d):
                cl = int(vs._c[d, l])
                if cl == 1:
                    dot = T(dot + uu[d])
                elif cl == -1:
                    dot = T(dot - uu[d])
            cu = T(T(3.0) * dot)
            out[l] = T(T(rho * w[l]) * T(T(T(1.0) + T(cu * T(T(1.0) + T(T(0.5) * cu)))) - usqr))
        return out
