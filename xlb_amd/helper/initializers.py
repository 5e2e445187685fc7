"""Equilibrium initialisation (reference xlb/helper/initializers.py:25-72): f = feq(rho, u),
defaults rho = 1, u = 0, evaluated in the compute dtype and written in f's dtype."""

from ..operator.equilibrium import QuadraticEquilibrium


def initialize_eq(f, grid, velocity_set, precision_policy, compute_backend, rho=None, u=None):
    if rho is None:
        rho = grid.create_field(cardinality=1, fill_value=1.0, dtype=precision_policy.compute_precision)
    if u is None:
        u = grid.create_field(cardinality=velocity_set.d, fill_value=0.0, dtype=precision_policy.compute_precision)
    equilibrium = QuadraticEquilibrium(velocity_set, precision_policy, compute_backend)
    return equilibrium(rho, u, f)
