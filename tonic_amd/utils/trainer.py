"""Training loop with the contract of ``tonic/utils/trainer.py:9-146`` — constructor arguments,
``initialize`` / ``run``, the ``train/*`` and ``test/*`` log keys, the checkpoint cadence and
the ``steps_per_second`` definition (epoch steps / epoch wall time) — built for large worker
counts: the per-step work on the host is a handful of NumPy calls on ``[W]`` arrays instead of
a Python loop over workers (``trainer.py:64-71`` is O(W) interpreter work per step, 256 to
10 240 iterations at the benchmark sizes).

The loop body is three collaborators:

  * ``EpisodeLedger``  running return / length of every worker; finished episodes are cut out
                       with one boolean mask per step and logged in worker order;
  * ``EpochClock``     counts environment steps towards the next epoch report and the next
                       checkpoint, and owns the wall-clock bookkeeping of the report;
  * ``CheckpointKeeper`` where checkpoints go and which older ones are dropped.

The reference's own ``tonic.Trainer`` drives the agents of this package just as well (they are
duck-typed against ``tonic/agents/agent.py``); this class exists for ``import tonic_amd as
tonic`` and for the vectorised bookkeeping.
"""
import os
import time

import numpy as np

from tonic_amd.utils import logger


class EpisodeLedger:
    """Return and length of the running episode of every worker."""

    def __init__(self, workers, prefix):
        self.returns = np.zeros(workers)
        self.lengths = np.zeros(workers, int)
        self.prefix = prefix
        self.finished = 0

    def advance(self, rewards, resets):
        """Accounts one environment step; logs and clears the episodes that just ended."""
        self.returns += rewards
        self.lengths += 1
        ended = np.asarray(resets, bool)
        if not ended.any():
            return
        for value in self.returns[ended]:                 # worker order, like the reference
            logger.store(self.prefix + '/episode_score', value, stats=True)
        for value in self.lengths[ended]:
            logger.store(self.prefix + '/episode_length', value, stats=True)
        self.returns[ended] = 0
        self.lengths[ended] = 0
        self.finished += int(ended.sum())


class EpochClock:
    """Environment-step counters of the run and the numbers of one epoch report."""

    def __init__(self, workers, epoch_steps, save_steps, max_steps):
        self.workers = workers
        self.epoch_length, self.save_interval, self.max_steps = epoch_steps, save_steps, max_steps
        self.total = self.in_epoch = self.since_save = self.epochs = 0
        self.started = self.epoch_started = time.time()

    def tick(self):
        self.total += self.workers
        self.in_epoch += self.workers
        self.since_save += self.workers

    @property
    def epoch_over(self):
        return self.in_epoch >= self.epoch_length

    @property
    def finished(self):
        return self.total >= self.max_steps

    @property
    def checkpoint_due(self):
        return self.finished or self.since_save >= self.save_interval

    def report(self, episodes):
        """Stores the epoch summary and starts the next epoch (trainer.py:78-94)."""
        self.epochs += 1
        now = time.time()
        elapsed = now - self.epoch_started
        for key, value in (('episodes', episodes), ('epochs', self.epochs),
                           ('seconds', now - self.started), ('epoch_seconds', elapsed),
                           ('epoch_steps', self.in_epoch), ('steps', self.total),
                           ('worker_steps', self.total // self.workers),
                           ('steps_per_second', self.in_epoch / elapsed)):
            logger.store('train/' + key, value)
        logger.dump()
        self.epoch_started = time.time()
        self.in_epoch = 0

    def saved(self):
        self.since_save = self.total % self.save_interval


class CheckpointKeeper:
    """``<experiment path>/checkpoints/step_<steps>``; optionally only the newest one."""

    def __init__(self, keep_only_last):
        self.keep_only_last = keep_only_last

    def save(self, agent, steps):
        folder = os.path.join(logger.get_path(), 'checkpoints')
        if self.keep_only_last and os.path.isdir(folder):
            for name in os.listdir(folder):
                if name.startswith('step_'):
                    os.remove(os.path.join(folder, name))
        agent.save(os.path.join(folder, f'step_{steps}'))


class Trainer:
    def __init__(self, steps=int(1e7), epoch_steps=int(2e4), save_steps=int(5e5),
                 test_episodes=5, show_progress=True, replace_checkpoint=False):
        self.max_steps = steps
        self.epoch_steps = epoch_steps
        self.save_steps = save_steps
        self.test_episodes = test_episodes
        self.show_progress = show_progress
        self.replace_checkpoint = replace_checkpoint

    def initialize(self, agent, environment, test_environment=None):
        self.agent = agent
        self.environment = environment
        self.test_environment = test_environment

    def run(self):
        agent, environment = self.agent, self.environment
        from tonic_amd import parallel
        rank, world = parallel.launch_rank()
        observations = environment.start()
        workers = len(observations)
        # `steps` counts the environment steps of the whole job (every rank steps its share of the
        # workers in lockstep): schedules written in steps — epochs, checkpoints, Buffer.ready,
        # exploration start — mean the same thing on one GPU and on eight
        clock = EpochClock(workers * world, self.epoch_steps, self.save_steps, self.max_steps)
        ledger = EpisodeLedger(workers, 'train')
        keeper = CheckpointKeeper(self.replace_checkpoint)
        self.steps = 0
        while not clock.finished:
            actions = agent.step(observations, self.steps)
            if np.isnan(actions.sum()):
                raise AssertionError('the agent produced NaN actions')
            logger.store('train/action', actions, stats=True)
            observations, infos = environment.step(actions)
            agent.update(**infos, steps=self.steps)
            clock.tick()
            self.steps = clock.total
            ledger.advance(infos['rewards'], infos['resets'])
            if self.show_progress:
                logger.show_progress(self.steps, self.epoch_steps, self.max_steps)
            if clock.epoch_over:
                if self.test_environment:
                    self._test()
                clock.report(ledger.finished)
            if clock.checkpoint_due:
                if rank == 0:                          # replicated parameters: one copy is enough
                    keeper.save(agent, self.steps)
                clock.saved()

    def _test(self):
        """``test_episodes`` whole episodes on the single-worker test environment; its
        observation survives from one epoch to the next (trainer.py:114-146)."""
        if not hasattr(self, 'test_observations'):
            self.test_observations = self.test_environment.start()
            if len(self.test_observations) != 1:
                raise AssertionError('the test environment must have exactly one worker')
        ledger = EpisodeLedger(1, 'test')
        while ledger.finished < self.test_episodes:
            actions = self.agent.test_step(self.test_observations, self.steps)
            if np.isnan(actions.sum()):
                raise AssertionError('the agent produced NaN test actions')
            logger.store('test/action', actions, stats=True)
            self.test_observations, infos = self.test_environment.step(actions)
            self.agent.test_update(**infos, steps=self.steps)
            ledger.advance(infos['rewards'], infos['resets'])
