"""tonic_amd — MI355X-native rollout-collect + learner-update engine behind Tonic's API.

Import it the way ``tonic`` is imported; the sub-packages mirror the reference's names::

    import tonic_amd as tonic
    import tonic_amd.torch
    agent = tonic.torch.agents.PPO()
    environment = tonic.environments.distribute(builder, parallel, sequential)
    trainer = tonic.Trainer()
"""
from . import agents
from . import environments
from . import explorations
from .utils import logger
from . import replays
from .utils.trainer import Trainer

__all__ = ['agents', 'environments', 'explorations', 'logger', 'replays', 'Trainer']
__version__ = '0.1.0'
