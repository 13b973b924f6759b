from .distributed import distribute, Parallel, Sequential
from .synthetic import Box, Synthetic, SyntheticBatch

__all__ = ['distribute', 'Parallel', 'Sequential', 'Box', 'Synthetic', 'SyntheticBatch']
