"""The engine as a reinforcement-learning system: every agent, driven by the Trainer through
``distribute`` / ``step`` / ``update`` exactly as ``python -m tonic.train`` would, must LEARN a task
whose optimal action depends on the observation (parity tests pin single updates on the reference;
this pins that the pieces compose: collector, replay, returns, updates, normaliser, target networks).

Task: observations ~ N(0, 1)^O, reward = 1 - mean((a - tanh(M obs))^2) per step for a fixed random
M, episodes of 20 steps.  A policy that ignores the observation cannot get past ~0.6 with a = 0; a
uniformly random one sits near 0.1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

O, A = 6, 3


class Reach:
    def __init__(self):
        from tonic_amd.environments import Box
        self.observation_space = Box(-np.inf, np.inf, (O,))
        self.action_space = Box(-1, 1, (A,))
        self.max_episode_steps = 20
        self.name = 'reach'
        self.matrix = np.random.RandomState(7).standard_normal((A, O)).astype(np.float32) * 0.8
        self.random = np.random.RandomState(0)

    def seed(self, seed):
        self.random = np.random.RandomState(seed)

    def reset(self):
        self.observation = self.random.standard_normal(O).astype(np.float32)
        return self.observation

    def step(self, action):
        target = np.tanh(self.matrix @ self.observation)
        reward = 1.0 - float(np.mean(np.square(np.clip(action, -1, 1) - target)))
        Reach.rewards.append(reward)
        return self.reset(), reward, False, {}


def run(agent, steps, workers, tmp_path):
    import tonic_amd
    from tonic_amd import environments, logger
    logger.initialize(path=str(tmp_path))
    Reach.rewards = []
    env = environments.distribute(Reach, 1, workers)
    env.initialize(seed=1)
    agent.initialize(env.observation_space, env.action_space, seed=5)
    trainer = tonic_amd.Trainer(steps=steps, epoch_steps=steps, save_steps=10 * steps,
                                show_progress=False)
    trainer.initialize(agent, env)
    trainer.run()
    rewards = np.array(Reach.rewards)
    tenth = len(rewards) // 10
    return rewards[:tenth].mean(), rewards[-tenth:].mean()


# measured (first tenth -> last tenth of the training rewards, one MI355X): PPO 0.18 -> 0.79, TRPO 0.22 ->
# 0.91, A2C 0.09 -> 0.35 (one actor step per update), DDPG 0.32 -> 0.98, TD3 0.31 -> 0.98, SAC 0.17 ->
# 0.74, D4PG 0.22 -> 0.96, MPO 0.11 -> 0.49 (samples its actions, lr 3e-4)
@pytest.mark.parametrize('name,floor', [('PPO', 0.7), ('A2C', 0.25), ('TRPO', 0.8)])
def test_on_policy_agents_learn(tmp_path, name, floor):
    import tonic_amd
    import tonic_amd.torch
    agent = getattr(tonic_amd.torch.agents, name)(
        replay=tonic_amd.replays.Segment(size=64, batch_iterations=20))
    first, last = run(agent, steps=64 * 16 * 40, workers=16, tmp_path=tmp_path)
    assert last > first + 0.15 and last > floor, (name, first, last)


@pytest.mark.parametrize('name,kwargs,floor', [
    ('DDPG', {}, 0.9), ('TD3', {}, 0.9), ('SAC', {}, 0.65), ('D4PG', dict(return_steps=3), 0.85),
    ('MPO', dict(return_steps=3), 0.4)])
def test_off_policy_agents_learn(tmp_path, name, kwargs, floor):
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import explorations
    replay = tonic_amd.replays.Buffer(size=20000, batch_iterations=20, batch_size=100,
                                      discount_factor=0.9, steps_before_batches=400,
                                      steps_between_batches=40, **kwargs)
    cls = getattr(tonic_amd.torch.agents, name)
    if name == 'MPO':
        agent = cls(replay=replay)
    elif name == 'SAC':
        agent = cls(replay=replay, exploration=explorations.NoActionNoise(start_steps=400))
    else:
        agent = cls(replay=replay, exploration=explorations.NormalActionNoise(start_steps=400))
    if name == 'D4PG':                      # a support for this task's values: (1 - mse) / (1 - 0.9)
        from tonic_amd.torch import models
        agent.model.critic.head = models.DistributionalValueHead(-2., 12., 51)
    first, last = run(agent, steps=6000, workers=4, tmp_path=tmp_path)
    assert last > first + 0.15 and last > floor, (name, first, last)
