"""Agent interface — ``tonic/agents/agent.py:4-34`` (the duck-typed boundary the Trainer calls)."""
import abc

import numpy as np


class Agent(abc.ABC):
    """Abstract class used to build agents."""

    def initialize(self, observation_space, action_space, seed=None):
        pass

    @abc.abstractmethod
    def step(self, observations, steps):
        """Returns actions during training."""

    def update(self, observations, rewards, resets, terminations, steps):
        """Informs the agent of the latest transitions during training."""

    @abc.abstractmethod
    def test_step(self, observations, steps):
        """Returns actions during testing."""

    def test_update(self, observations, rewards, resets, terminations, steps):
        """Informs the agent of the latest transitions during testing."""

    def save(self, path):
        """Saves the agent weights during training."""

    def load(self, path):
        """Reloads the agent weights from a checkpoint."""


class _Scripted(Agent):
    """The non-learning agents of ``tonic/agents/basic.py`` (debugging aids: pure host code, no
    model, nothing for the GPU): same constructor arguments, same NumPy ``RandomState`` stream."""

    def initialize(self, observation_space, action_space, seed=None):
        self.action_size = action_space.shape[0]
        self.np_random = np.random.RandomState(seed)

    def step(self, observations, steps):
        return self._policy(observations)

    def test_step(self, observations, steps):
        return self._policy(observations)


class NormalRandom(_Scripted):
    """basic.py:8-28."""

    def __init__(self, loc=0, scale=1):
        self.loc, self.scale = loc, scale

    def _policy(self, observations):
        return self.np_random.normal(self.loc, self.scale, (len(observations), self.action_size))


class UniformRandom(_Scripted):
    """basic.py:31-47."""

    def _policy(self, observations):
        return self.np_random.uniform(-1, 1, (len(observations), self.action_size))


class Constant(_Scripted):
    """basic.py:103-119."""

    def __init__(self, constant=0):
        self.constant = constant

    def _policy(self, observations):
        return np.full((len(observations), self.action_size), self.constant)


class OrnsteinUhlenbeck(_Scripted):
    """basic.py:50-100: two independent OU processes (training / test workers), each zeroed for
    the workers whose episode was reset."""

    def __init__(self, scale=0.2, clip=2, theta=.15, dt=1e-2):
        self.scale, self.clip, self.theta, self.dt = scale, clip, theta, dt

    def initialize(self, observation_space, action_space, seed=None):
        super().initialize(observation_space, action_space, seed)
        self.train_actions = self.test_actions = None

    def _advance(self, actions, observations):
        if actions is None:
            actions = np.zeros((len(observations), self.action_size))
        draws = np.clip(self.np_random.normal(size=actions.shape), -self.clip, self.clip)
        moved = (1 - self.theta * self.dt) * actions
        moved += self.scale * np.sqrt(self.dt) * draws
        return np.clip(moved, -1, 1)

    def step(self, observations, steps):
        self.train_actions = self._advance(self.train_actions, observations)
        return self.train_actions

    def test_step(self, observations, steps):
        self.test_actions = self._advance(self.test_actions, observations)
        return self.test_actions

    def update(self, observations, rewards, resets, terminations, steps):
        self.train_actions *= (1. - resets)[:, None]

    def test_update(self, observations, rewards, resets, terminations, steps):
        self.test_actions *= (1. - resets)[:, None]


__all__ = ['Agent', 'Constant', 'NormalRandom', 'OrnsteinUhlenbeck', 'UniformRandom']
