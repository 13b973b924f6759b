"""Action-noise explorations — API of ``tonic/explorations/noisy.py`` (host logic: the noise
must come from the same NumPy ``RandomState`` stream as the reference's)."""
import numpy as np


class NoActionNoise:
    """noisy.py:6-25 (SAC): uniform warm-up actions (float64, quirk Q10), then the policy."""

    def __init__(self, start_steps=20000):
        self.start_steps = start_steps

    def initialize(self, policy, action_space, seed=None):
        from tonic_amd import parallel
        self.policy = policy
        self.action_size = action_space.shape[0]
        self.np_random = np.random.RandomState(seed)
        # several ranks: the draws of ALL workers with TONIC_AMD_GLOBAL_NOISE=1 (this rank keeps the
        # rows of its own: the single-process stream), else a stream of this rank's own
        self.rank, self.world = parallel.launch_rank()
        self.global_noise = self.world > 1 and parallel.global_noise()
        if self.world > 1 and not self.global_noise and seed is not None:
            self.np_random = np.random.RandomState(seed + self.rank)

    def _draw(self, method, workers, *args):
        """`method(*args, size=(workers, action_size))` for this rank's workers."""
        if not self.global_noise:
            return method(*args, size=(workers, self.action_size))
        rows = method(*args, size=(self.world * workers, self.action_size))
        return rows[self.rank * workers:(self.rank + 1) * workers]

    def _warm_up(self, observations):
        return self._draw(self.np_random.uniform, len(observations), -1, 1)

    def __call__(self, observations, steps):
        if steps > self.start_steps:
            return np.clip(self.policy(observations), -1, 1)
        return self._warm_up(observations)

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
            noises = self.scale * self._draw(self.np_random.normal, len(actions))
            return np.clip((actions + noises).astype(np.float32), -1, 1)
        return self._warm_up(observations)


class OrnsteinUhlenbeckActionNoise(NoActionNoise):
    """noisy.py:53-91: policy + a per-worker Ornstein-Uhlenbeck state (mean reversion theta * dt,
    clipped N(0, 1) increments scaled by scale * sqrt(dt)) that `update` zeroes for the workers
    whose episode was reset."""

    def __init__(self, scale=0.1, clip=2, theta=.15, dt=1e-2, start_steps=20000):
        super().__init__(start_steps)
        self.scale, self.clip, self.theta, self.dt = scale, clip, theta, dt

    def initialize(self, policy, action_space, seed=None):
        super().initialize(policy, action_space, seed)
        self.noises = None

    def __call__(self, observations, steps):
        if steps <= self.start_steps:
            return self._warm_up(observations)
        actions = self.policy(observations)
        if self.noises is None:
            self.noises = np.zeros_like(actions)
        draws = np.clip(self._draw(self.np_random.normal, len(actions)), -self.clip, self.clip)
        self.noises -= self.theta * self.noises * self.dt
        self.noises += self.scale * np.sqrt(self.dt) * draws
        return np.clip((actions + self.noises).astype(np.float32), -1, 1)

    def update(self, resets):
        if self.noises is not None:
            self.noises *= (1. - resets)[:, None]


__all__ = ['NoActionNoise', 'NormalActionNoise', 'OrnsteinUhlenbeckActionNoise']
