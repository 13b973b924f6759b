"""Training loop — API and logged keys of ``tonic/utils/trainer.py:9-146``.

Same constructor arguments, ``initialize`` / ``run`` protocol, ``train/*`` and ``test/*`` keys,
checkpoint cadence and ``steps_per_second`` definition (epoch steps / epoch wall time,
trainer.py:81-91).  The per-step O(W) Python loop over workers (trainer.py:64-71) is
vectorised with NumPy: episode scores/lengths are logged in worker order, as before.
"""
import os
import time

import numpy as np

from tonic_amd.utils import logger


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
        start_time = last_epoch_time = time.time()
        observations = self.environment.start()
        num_workers = len(observations)
        scores = np.zeros(num_workers)
        lengths = np.zeros(num_workers, int)
        self.steps, epoch_steps, epochs, episodes, steps_since_save = 0, 0, 0, 0, 0

        while True:
            actions = self.agent.step(observations, self.steps)
            assert not np.isnan(actions.sum())
            logger.store('train/action', actions, stats=True)

            observations, infos = self.environment.step(actions)
            self.agent.update(**infos, steps=self.steps)

            scores += infos['rewards']
            lengths += 1
            self.steps += num_workers
            epoch_steps += num_workers
            steps_since_save += num_workers

            if self.show_progress:
                logger.show_progress(self.steps, self.epoch_steps, self.max_steps)

            finished = np.flatnonzero(infos['resets'])
            if finished.size:
                for i in finished:
                    logger.store('train/episode_score', scores[i], stats=True)
                    logger.store('train/episode_length', lengths[i], stats=True)
                scores[finished] = 0
                lengths[finished] = 0
                episodes += finished.size

            if epoch_steps >= self.epoch_steps:
                if self.test_environment:
                    self._test()
                epochs += 1
                now = time.time()
                epoch_time = now - last_epoch_time
                logger.store('train/episodes', episodes)
                logger.store('train/epochs', epochs)
                logger.store('train/seconds', now - start_time)
                logger.store('train/epoch_seconds', epoch_time)
                logger.store('train/epoch_steps', epoch_steps)
                logger.store('train/steps', self.steps)
                logger.store('train/worker_steps', self.steps // num_workers)
                logger.store('train/steps_per_second', epoch_steps / epoch_time)
                logger.dump()
                last_epoch_time = time.time()
                epoch_steps = 0

            stop_training = self.steps >= self.max_steps
            if stop_training or steps_since_save >= self.save_steps:
                path = os.path.join(logger.get_path(), 'checkpoints')
                if os.path.isdir(path) and self.replace_checkpoint:
                    for name in os.listdir(path):
                        if name.startswith('step_'):
                            os.remove(os.path.join(path, name))
                self.agent.save(os.path.join(path, f'step_{self.steps}'))
                steps_since_save = self.steps % self.save_steps
            if stop_training:
                break

    def _test(self):
        """trainer.py:114-146: `test_episodes` episodes on the single-worker test environment;
        the observation is kept across epochs."""
        if not hasattr(self, 'test_observations'):
            self.test_observations = self.test_environment.start()
            assert len(self.test_observations) == 1
        for _ in range(self.test_episodes):
            score, length = 0, 0
            while True:
                actions = self.agent.test_step(self.test_observations, self.steps)
                assert not np.isnan(actions.sum())
                logger.store('test/action', actions, stats=True)
                self.test_observations, infos = self.test_environment.step(actions)
                self.agent.test_update(**infos, steps=self.steps)
                score += infos['rewards'][0]
                length += 1
                if infos['resets'][0]:
                    break
            logger.store('test/episode_score', score, stats=True)
            logger.store('test/episode_length', length, stats=True)
