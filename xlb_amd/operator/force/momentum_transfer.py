"""Momentum-exchange force on a solid (reference xlb/operator/force/momentum_transfer.py:121-205, JAX semantics,
stream-then-collide sequence): ``MomentumTransfer(no_slip_bc)(f_0, f_1, bc_mask, missing_mask) -> force (d,)``.

``f_0`` is what a step returned (post-collision).  At every cell of the no-slip BC that is not solid itself, each
missing direction l contributes ``c_opp(l) * (f_0[opp l] + f_post_stream[l])``, where the post-stream value is the
BC's bounce-back of the own cell — so neither a streamed copy of the field nor ``f_1`` is needed (``f_1`` is accepted
for API compatibility, as in the reference).  The sum over the grid is accumulated in double precision on the device;
under slab decomposition the ranks' vectors are added with ``xlb_amd.distribute.all_reduce_sum``."""

import ctypes as C

import numpy as np

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class MomentumTransfer(Operator):
    def __init__(self, no_slip_bc_instance, velocity_set=None, precision_policy=None, compute_backend=None):
        self.no_slip_bc_instance = no_slip_bc_instance
        super().__init__(velocity_set, precision_policy, compute_backend)
        self._via_stepper = no_slip_bc_instance.hip_kind in (_lib.BC_HYBRID_BB_REGULARIZED, _lib.BC_HYBRID_BB_GRADS, _lib.BC_HYBRID_NEQ_REGULARIZED,
                                                             _lib.BC_HALFWAY_BB_PROFILE)
        if not self._via_stepper and no_slip_bc_instance.hip_kind not in (_lib.BC_HALFWAY_BB, _lib.BC_FULLWAY_BB):
            raise NotImplementedError("MomentumTransfer supports halfway / fullway bounce-back walls and HybridBC on the HIP backend")

    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_0, f_1, bc_mask, missing_mask):
        if self._via_stepper:
            # HybridBC / profile walls: the BC's post-stream populations need the wall-distance and wall-velocity tables, which
            # live in the stepper the BC belongs to (the reference applies bc.warp_functional here, momentum_transfer.py:75-92)
            owner = getattr(self.no_slip_bc_instance, "_stepper_ref", None)
            stepper = owner() if owner is not None else None
            if stepper is None:
                raise RuntimeError("MomentumTransfer(HybridBC): the BC is not part of a live stepper (stepper.prepare_fields() comes first)")
            force = stepper._native_stepper().momentum_transfer(self.no_slip_bc_instance.id, f_0, bc_mask, missing_mask)
            return force[3 - self.velocity_set.d :].astype(self.compute_dtype)
        desc = self.no_slip_bc_instance._hip_descriptor()
        out = (C.c_double * 3)()
        _lib.check(
            _lib.load().xlbhip_momentum_transfer(
                self._ctx.handle, self.velocity_set.hip_id, self._compute_code, C.byref(desc), f_0.handle, bc_mask.handle, missing_mask.handle, out
            )
        )
        force = np.array(out[:], dtype=np.float64)
        if f_0.halo > 0:
            from ...distribute import all_reduce_sum

            force = np.array([all_reduce_sum(v) for v in force])
        return force[3 - self.velocity_set.d :].astype(self.compute_dtype)
