"""Fixed-shape synthetic environments used for benchmarks and tests: gym,
pybullet and dm_control are not installed in the ROCm image, and the benchmark contract asks
for synthetic observation / action batches of the configured shapes anyway.

``Synthetic`` is one zero-cost environment with the attributes ``distribute`` touches
(distributed.py:12-20,24,36,47,161).  ``SyntheticBatch`` is the vectorised form: W workers
stepped by one NumPy call, same ``start`` / ``step`` protocol as ``Sequential``.
"""
import numpy as np


class Box:
    """The subset of gym.spaces.Box the agents read (.shape/.low/.high/.dtype)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            low, high = np.full(shape, low, dtype), np.full(shape, high, dtype)
        self.low, self.high = np.asarray(low, dtype), np.asarray(high, dtype)
        self.shape, self.dtype = self.low.shape, np.dtype(dtype)


class Synthetic:
    """obs ~ N(0,1) float32, reward = -||a||^2, never terminates (time-outs only)."""

    def __init__(self, observation_size, action_size, max_episode_steps=1000, name=None):
        self.observation_space = Box(-np.inf, np.inf, (observation_size,))
        self.action_space = Box(-1, 1, (action_size,))
        self.max_episode_steps = max_episode_steps
        self.name = name or f'synthetic-{observation_size}-{action_size}'
        self.random = np.random.RandomState(0)

    def seed(self, seed):
        self.random = np.random.RandomState(seed)

    def reset(self):
        return self.random.normal(size=self.observation_space.shape).astype(np.float32)

    def step(self, action):
        observation = self.random.normal(size=self.observation_space.shape).astype(np.float32)
        return observation, -float(np.sum(np.square(action))), False, {}


class SyntheticBatch:
    """W synthetic workers stepped at once (one RandomState for the whole batch)."""

    def __init__(self, workers, observation_size, action_size, max_episode_steps=1000,
                 termination_probability=0.0, name=None):
        self.workers = workers
        self.observation_space = Box(-np.inf, np.inf, (observation_size,))
        self.action_space = Box(-1, 1, (action_size,))
        self.max_episode_steps = max_episode_steps
        self.termination_probability = termination_probability
        self.name = name or f'synthetic-{observation_size}-{action_size}'

    def initialize(self, seed):
        self.random = np.random.RandomState(seed)

    def _observe(self):
        shape = (self.workers,) + self.observation_space.shape
        return self.random.standard_normal(shape).astype(np.float32)

    def start(self):
        self.lengths = np.zeros(self.workers, int)
        self.observations = self._observe()
        return self.observations.copy()

    def step(self, actions):
        next_observations = self._observe()
        rewards = -np.square(np.asarray(actions, np.float32)).sum(-1)
        self.lengths += 1
        if self.termination_probability > 0:
            terminations = self.random.uniform(size=self.workers) < self.termination_probability
        else:
            terminations = np.zeros(self.workers, bool)
        resets = terminations | (self.lengths == self.max_episode_steps)
        observations = next_observations.copy()
        if resets.any():
            observations[resets] = self._observe()[resets]
            self.lengths[resets] = 0
        infos = dict(observations=next_observations, rewards=rewards.astype(np.float32),
                     resets=resets, terminations=terminations)
        return observations, infos
