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
    """W synthetic workers stepped at once (one RandomState for the whole batch), writing into a
    collector block like ``Sequential`` / ``Parallel`` do and handing out its views.

    ``pool`` > 0 pre-generates that many observation batches and cycles through them, so that a
    step costs two row copies and the reward instead of W*O fresh normal draws: the simulator
    then costs (almost) nothing and a loop over it measures the collector and the agent."""

    def __init__(self, workers, observation_size, action_size, max_episode_steps=1000,
                 termination_probability=0.0, name=None, pool=0, copy_outputs=False):
        self.workers = workers
        self.observation_space = Box(-np.inf, np.inf, (observation_size,))
        self.action_space = Box(-1, 1, (action_size,))
        self.max_episode_steps = max_episode_steps
        self.termination_probability = termination_probability
        self.name = name or f'synthetic-{observation_size}-{action_size}'
        self.pool = pool
        self.copy_outputs = copy_outputs

    def initialize(self, seed):
        self.random = np.random.RandomState(seed)
        if self.pool:
            shape = (self.pool, self.workers) + self.observation_space.shape
            self._pool = self.random.standard_normal(shape).astype(np.float32)
            self._cursor = 0

    def _observe(self):
        if self.pool:
            self._cursor = (self._cursor + 1) % self.pool
            return self._pool[self._cursor]
        shape = (self.workers,) + self.observation_space.shape
        return self.random.standard_normal(shape).astype(np.float32)

    def start(self):
        from tonic_amd import _lib
        from tonic_amd.collector import Block
        self.block = Block(self.workers, self.observation_space.shape[0],
                           self.action_space.shape[0])
        self.block.promise_carry_over()      # (step: observations = next observations | reset rows)
        self.lengths = np.zeros(self.workers, int)
        self._longest = 0                # max(lengths), kept by hand: a reduction per step is ~1 us
        self._flags = np.zeros(self.workers, bool)
        self._quiet = 0                  # steps taken since `lengths` was last brought up to date
        self._flags_set = False
        self.block.observations[:] = self._observe()
        if self.pool:
            self._synthetic_step = _lib.hot('tonic_collector_synthetic_step')
            self._block_address = self.block.address
            self._pool_rows = [row.ctypes.data for row in self._pool]      # (bound once: hot path)
        return self.block.observations.copy() if self.copy_outputs else self.block.out_observations

    def _outputs(self):
        block = self.block
        if self.copy_outputs:
            return block.observations.copy(), {k: v.copy() for k, v in block.infos.items()}
        return block.out_observations, dict(block.infos)

    def _write_flags(self, resets, terminations):
        block = self.block
        np.copyto(block.resets, resets)
        np.copyto(block.terminations, terminations)
        np.copyto(block.resets_bool, resets)
        np.copyto(block.terminations_bool, terminations)

    def step(self, actions):
        block = self.block
        # nobody can time out at this step: the flags stay all False
        quiet = self.termination_probability <= 0 and \
            self._quiet + 1 + self._longest < self.max_episode_steps
        rung = False
        if self.pool and actions is block.out_actions:
            # the agent handed the block's own actions over: the whole record is one host call
            # (tonic_collector_synthetic_step) instead of five NumPy calls — and when the flags in
            # the block are final too, the record is complete inside that call, which then issues
            # the command the agent has armed for this moment (Block.ring)
            rung = quiet and not self._flags_set
            self._cursor = (self._cursor + 1) % self.pool
            self._synthetic_step(self._block_address, self._pool_rows[self._cursor], None, rung)
        else:
            next_observations = self._observe()
            np.copyto(block.next_observations, next_observations)
            np.copyto(block.observations, next_observations)
            np.einsum('ij,ij->i', actions, actions, out=block.rewards, casting='same_kind')
            np.negative(block.rewards, out=block.rewards)
        if quiet:
            self._quiet += 1
            if self._flags_set:
                self._write_flags(self._flags, self._flags)
                self._flags_set = False
            if not rung:
                block.ring()
            return self._outputs()
        self.lengths += self._quiet + 1
        self._quiet = 0
        if self.termination_probability > 0:
            terminations = self.random.uniform(size=self.workers) < self.termination_probability
        else:
            terminations = self._flags                          # all False
        resets = terminations | (self.lengths == self.max_episode_steps)
        if resets.any():
            block.observations[resets] = self._observe()[resets]
            self.lengths[resets] = 0
        self._longest = int(self.lengths.max())
        self._write_flags(resets, terminations)
        self._flags_set = True
        block.ring()
        return self._outputs()
