"""Post-processing operators of the HIP backend (reference xlb/operator/postprocess/)."""

from .postprocess import Vorticity as Vorticity, QCriterion as QCriterion, GridToPoint as GridToPoint
