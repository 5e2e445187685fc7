"""Force operators of the HIP backend (reference xlb/operator/force/)."""

from .momentum_transfer import MomentumTransfer as MomentumTransfer
