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



def install(collector=True, trainer=False):
    """For runs launched by the REFERENCE's ``python -m tonic.train``: that script resolves
    ``tonic.environments.distribute`` and ``tonic.Trainer`` from its own module globals
    (tonic/train.py:83-91,120-121), which a ``--header`` cannot rebind — but it can call this:

        --header 'import tonic_amd, tonic_amd.torch; tonic_amd.install()'

    makes ``tonic.environments.distribute`` the shared-block collector of this package (the
    agents then read the workers' memory in place) and, with ``trainer=True``, ``tonic.Trainer``
    the vectorised trainer."""
    import sys
    reference = sys.modules.get('tonic')
    if reference is None:
        import tonic as reference
    if collector:
        reference.environments.distribute = environments.distribute
    if trainer:
        reference.Trainer = Trainer
    return reference


__all__ = ['agents', 'environments', 'explorations', 'install', 'logger', 'replays', 'Trainer']
__version__ = '0.1.0'
