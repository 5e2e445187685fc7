from .stream import Stream as Stream
