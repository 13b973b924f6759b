from .buffers import Buffer
from .segments import Segment, flatten_batch

__all__ = ['Buffer', 'Segment', 'flatten_batch']
