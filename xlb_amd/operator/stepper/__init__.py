"""Steppers of the HIP backend: the fused incompressible Navier-Stokes step."""

from .stepper import Stepper as Stepper
from .nse_stepper import IncompressibleNavierStokesStepper as IncompressibleNavierStokesStepper
