from . import models, normalizers, updaters
from . import agents

__all__ = ['agents', 'models', 'normalizers', 'updaters']
