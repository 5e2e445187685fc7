from .quadratic_equilibrium import Equilibrium as Equilibrium, QuadraticEquilibrium as QuadraticEquilibrium
