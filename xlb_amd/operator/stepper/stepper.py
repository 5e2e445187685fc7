"""Stepper base (reference xlb/operator/stepper/stepper.py:6-34)."""

from ...default_config import DefaultConfig
from ..operator import Operator


class Stepper(Operator):
    def __init__(self, grid, boundary_conditions=[]):
        self.grid = grid
        self.boundary_conditions = list(boundary_conditions)
        velocity_sets = {bc.velocity_set for bc in self.boundary_conditions} or {DefaultConfig.velocity_set}
        policies = {bc.precision_policy for bc in self.boundary_conditions} or {DefaultConfig.default_precision_policy}
        backends = {bc.compute_backend for bc in self.boundary_conditions} or {DefaultConfig.default_backend}
        assert len(velocity_sets) == 1, "All velocity sets must be the same"
        assert len(policies) == 1, "All precision policies must be the same"
        assert len(backends) == 1, "All compute backends must be the same"
        super().__init__(velocity_sets.pop(), policies.pop(), backends.pop())
