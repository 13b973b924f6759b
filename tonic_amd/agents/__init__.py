"""Agent interface — ``tonic/agents/agent.py:4-34`` (the duck-typed boundary the Trainer calls)."""
import abc


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


__all__ = ['Agent']
