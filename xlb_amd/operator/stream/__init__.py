"""Stand-alone pull-streaming operator of the HIP backend (the fused stepper streams in-kernel)."""

from .stream import Stream as Stream
