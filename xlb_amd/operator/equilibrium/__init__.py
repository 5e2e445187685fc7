"""Equilibrium operators of the HIP backend."""

from .quadratic_equilibrium import Equilibrium as Equilibrium, QuadraticEquilibrium as QuadraticEquilibrium
