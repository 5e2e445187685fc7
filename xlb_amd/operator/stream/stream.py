"""Pull-scheme periodic streaming as a stand-alone operator
(reference xlb/operator/stream/stream.py:29-62 semantics, :114-125 call style)."""

from ... import _lib
from ...compute_backend import ComputeBackend
from ..operator import Operator


class Stream(Operator):
    @Operator.register_backend(ComputeBackend.HIP)
    def hip_implementation(self, f_0, f_1):
        _lib.check(_lib.load().xlbhip_stream(self._ctx.handle, self.velocity_set.hip_id, f_0.handle, f_1.handle))
        return f_1
