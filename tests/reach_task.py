"""The task of tests/test_gpu_learning.py and oracle/make_learning_curves.py (shared so that this
package's agents and the reference's are trained on exactly the same thing): observations ~
N(0, 1)^6, reward = 1 - mean((a - tanh(M obs))^2) for a fixed random M, time-outs after 20 steps."""
import numpy as np

O, A = 6, 3
ON_POLICY = dict(size=64, batch_iterations=20)                  # Segment
ON_POLICY_RUN = dict(steps=64 * 16 * 40, workers=16)
OFF_POLICY = dict(size=20000, batch_iterations=20, batch_size=100, discount_factor=0.9,
                  steps_before_batches=400, steps_between_batches=40)         # Buffer
OFF_POLICY_RUN = dict(steps=6000, workers=4)
START_STEPS = 400
D4PG_SUPPORT = (-2., 12., 51)              # this task's values: (1 - mse) / (1 - 0.9)
SEEDS = dict(environment=1, agent=5)


class _Space:
    def __init__(self, low, high, shape):
        self.low, self.high = np.full(shape, low, np.float32), np.full(shape, high, np.float32)
        self.shape, self.dtype = shape, np.dtype(np.float32)


class Reach:
    rewards = []

    def __init__(self):
        self.observation_space = _Space(-np.inf, np.inf, (O,))
        self.action_space = _Space(-1, 1, (A,))
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


def build_agent(tonic, torch_agents, name):
    """The agent `name` of either package (`tonic` = tonic_amd or the reference's tonic) in the
    configuration of this task."""
    if name in ('PPO', 'A2C', 'TRPO'):
        return getattr(torch_agents, name)(replay=tonic.replays.Segment(**ON_POLICY))
    extra = dict(return_steps=3) if name in ('D4PG', 'MPO') else {}
    replay = tonic.replays.Buffer(**OFF_POLICY, **extra)
    cls = getattr(torch_agents, name)
    if name == 'MPO':
        return cls(replay=replay)
    noise = tonic.explorations.NoActionNoise if name == 'SAC' else tonic.explorations.NormalActionNoise
    model = None
    if name == 'D4PG':          # d4pg.py:7-18 with a support for this task's values
        package = __import__(torch_agents.__name__.rsplit('.', 1)[0], fromlist=['models', 'normalizers'])
        import torch
        models = package.models
        model = models.ActorCriticWithTargets(
            actor=models.Actor(encoder=models.ObservationEncoder(),
                               torso=models.MLP((256, 256), torch.nn.ReLU),
                               head=models.DeterministicPolicyHead()),
            critic=models.Critic(encoder=models.ObservationActionEncoder(),
                                 torso=models.MLP((256, 256), torch.nn.ReLU),
                                 head=models.DistributionalValueHead(*D4PG_SUPPORT)),
            observation_normalizer=package.normalizers.MeanStd())
    return cls(model=model, replay=replay, exploration=noise(start_steps=START_STEPS))


def train(tonic, agent, name, path):
    """Runs the Trainer of `tonic`; returns the mean training reward of every tenth of the run."""
    run = ON_POLICY_RUN if name in ('PPO', 'A2C', 'TRPO') else OFF_POLICY_RUN
    tonic.logger.initialize(path=path)
    Reach.rewards = []
    environment = tonic.environments.distribute(Reach, 1, run['workers'])
    environment.initialize(seed=SEEDS['environment'])
    agent.initialize(environment.observation_space, environment.action_space, seed=SEEDS['agent'])
    trainer = tonic.Trainer(steps=run['steps'], epoch_steps=run['steps'], save_steps=10 * run['steps'],
                            show_progress=False)
    trainer.initialize(agent, environment)
    trainer.run()
    rewards = np.array(Reach.rewards)
    tenth = len(rewards) // 10
    return [float(rewards[i * tenth:(i + 1) * tenth].mean()) for i in range(10)]
