"""Action-noise explorations — API of ``tonic/explorations/noisy.py`` (host logic: the noise
must come from the same NumPy ``RandomState`` stream as the reference's)."""
import numpy as np


class NoActionNoise:
    """noisy.py:6-25 (SAC): uniform warm-up actions (float64, quirk Q10), then the policy."""

    def __init__(self, start_steps=20000):
        self.start_steps = start_steps

    def initialize(self, policy, action_space, seed=None):
        self.policy = policy
        self.action_size = action_space.shape[0]
        self.np_random = np.random.RandomState(seed)

    def __call__(self, observations, steps):
        if steps > self.start_steps:
            return np.clip(self.policy(observations), -1, 1)
        return self.np_random.uniform(-1, 1, (len(observations), self.action_size))

    def update(self, resets):
        pass


class NormalActionNoise(NoActionNoise):
    """noisy.py:28-50 (TD3 / DDPG): policy + scale * N(0, 1), float32, clipped."""

    def __init__(self, scale=0.1, start_steps=20000):
        super().__init__(start_steps)
        self.scale = scale

    def __call__(self, observations, steps):
        if steps > self.start_steps:
            actions = self.policy(observations)
            noises = self.scale * self.np_random.normal(size=actions.shape)
            return np.clip((actions + noises).astype(np.float32), -1, 1)
        return self.np_random.uniform(-1, 1, (len(observations), self.action_size))


__all__ = ['NoActionNoise', 'NormalActionNoise']
