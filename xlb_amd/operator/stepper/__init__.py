from .stepper import Stepper as Stepper
from .nse_stepper import IncompressibleNavierStokesStepper as IncompressibleNavierStokesStepper
