from .segments import Segment, flatten_batch

__all__ = ['Segment', 'flatten_batch']
