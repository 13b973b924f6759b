"""Test helper: an agent with the `tonic.agents.Agent` duck type that needs no GPU but logs and
checkpoints through the SAME tonic_amd code paths the HIP PPO agent uses (`log_ppo_update`,
`Agent.save`), so the drop-in plumbing can be exercised under the reference's own
`tonic.train.train()` in the CPU-only build container."""
import numpy as np

import tonic_amd
import tonic_amd.torch
from tonic_amd.torch import agents


class LoggingOnlyPPO(agents.Agent):
    def __init__(self, update_every=8, iterations=3):
        self.update_every, self.iterations = update_every, iterations
        self.model = agents.default_model()

    def initialize(self, observation_space, action_space, seed=None):
        super().initialize(seed=seed)
        self.model.initialize(observation_space, action_space)     # CPU parameters: save() works
        self.action_size = action_space.shape[0]
        self.random = np.random.RandomState(seed)
        self.calls = 0

    def step(self, observations, steps):
        from tonic_amd.collector import Block
        self.saw_block_views = Block.owner_of(observations) is not None
        return self.random.uniform(-1, 1, (len(observations), self.action_size)).astype(np.float32)

    def test_step(self, observations, steps):
        return self.random.uniform(-1, 1, (len(observations), self.action_size)).astype(np.float32)

    def update(self, observations, rewards, resets, terminations, steps):
        self.calls += 1
        if self.calls % self.update_every == 0:
            infos = np.zeros((2, self.iterations, 8), np.float32)
            infos[0, :, 6] = 1                                     # every actor iteration ran
            infos[:, :, 0] = self.random.uniform(size=(2, self.iterations))
            agents.log_ppo_update(infos)
