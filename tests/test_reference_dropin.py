"""The drop-in claims checked against the REFERENCE's own code in the build container (skipped
where /root/reference is absent, e.g. on the GPU box): `tonic.train.train()` drives an agent of
this package and the learner statistics land in the reference's log.csv; the reference's own
agent loads a checkpoint written by this package."""
import os
import sys

import numpy as np
import pytest

import reference_loader

pytestmark = pytest.mark.skipif(not reference_loader.reference_available(),
                                reason='needs the reference checkout (build container only)')

PPO_KEYS = ('actor/loss', 'actor/kl', 'actor/entropy', 'actor/clip_fraction', 'actor/std',
            'actor/stop', 'actor/iterations', 'critic/loss', 'critic/v', 'critic/iterations')


def test_reference_train_logs_this_packages_learner_keys(tmp_path, monkeypatch):
    """tonic/train.py:76-133 end to end with `--header 'import tonic_amd as amd'`: the agent logs
    through tonic_amd.logger, the reference initialised tonic.logger — one log.csv, one
    checkpoint folder, under the reference's environment/name/seed path."""
    tonic = reference_loader.load_reference()
    import tonic.train
    import tonic_amd
    from tonic_amd.utils import logger as amd_logger
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(amd_logger, 'current_logger', None)        # nobody initialised ours
    monkeypatch.setattr(tonic.logger, 'current_logger', None)
    tonic.train.train(
        header='import tonic_amd as amd, tonic_amd.torch, stub_agents',
        agent='stub_agents.LoggingOnlyPPO(update_every=8, iterations=3)',
        environment='__import__("tonic_amd").environments.Synthetic(5, 2, max_episode_steps=7)',
        test_environment=None,
        trainer='tonic.Trainer(steps=96, epoch_steps=48, save_steps=96, show_progress=False)',
        before_training=None, after_training=None, parallel=1, sequential=4, seed=3,
        name=None, environment_name=None, checkpoint='last', path=None)
    run = tmp_path / 'synthetic-5-2' / 'LoggingOnlyPPO-1x4' / '3'
    header = (run / 'log.csv').read_text().split('\n')[0].split(',')
    for key in PPO_KEYS + ('train/episode_score/mean', 'test/episode_score/mean',
                           'train/steps_per_second'):
        assert key in header, (key, header)
    assert (run / 'config.yaml').exists()
    assert (run / 'checkpoints' / 'step_96.pt').exists()
    assert amd_logger.current_logger is None, 'the reference logger must stay the only one'
    rows = (run / 'log.csv').read_text().strip().split('\n')
    assert len(rows) == 3                                           # header + two epochs


def test_this_packages_trainer_under_the_reference_logger(tmp_path, monkeypatch):
    """`--trainer 'amd.Trainer(...)'` with the reference's logger: same single log."""
    tonic = reference_loader.load_reference()
    import tonic.train
    from tonic_amd.utils import logger as amd_logger
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(amd_logger, 'current_logger', None)
    monkeypatch.setattr(tonic.logger, 'current_logger', None)
    tonic.train.train(
        header='import tonic_amd as amd, tonic_amd.torch, stub_agents',
        agent='stub_agents.LoggingOnlyPPO(update_every=8, iterations=3)',
        environment='__import__("tonic_amd").environments.Synthetic(5, 2, max_episode_steps=7)',
        test_environment=None,
        trainer='amd.Trainer(steps=96, epoch_steps=48, save_steps=96, show_progress=False)',
        before_training=None, after_training=None, parallel=1, sequential=4, seed=3,
        name='run', environment_name='env', checkpoint='last', path=None)
    header = (tmp_path / 'env' / 'run' / '3' / 'log.csv').read_text().split('\n')[0].split(',')
    for key in PPO_KEYS + ('train/episode_score/mean', 'test/episode_length/mean'):
        assert key in header, key
    assert (tmp_path / 'env' / 'run' / '3' / 'checkpoints' / 'step_96.pt').exists()


def test_install_puts_the_collector_under_the_reference_cli(tmp_path, monkeypatch):
    """`tonic_amd.install()` from the header: the reference's train() then builds its
    environments with this package's distribute (forked workers writing into the shared block)
    and the agent receives the block's views."""
    tonic = reference_loader.load_reference()
    import tonic.train
    import stub_agents
    from tonic_amd.utils import logger as amd_logger
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(amd_logger, 'current_logger', None)
    monkeypatch.setattr(tonic.logger, 'current_logger', None)
    monkeypatch.setattr(tonic.environments, 'distribute', tonic.environments.distribute)
    seen = []
    original = stub_agents.LoggingOnlyPPO.update

    def spy(self, *args, **kwargs):
        seen.append(self.saw_block_views)
        return original(self, *args, **kwargs)
    monkeypatch.setattr(stub_agents.LoggingOnlyPPO, 'update', spy)
    tonic.train.train(
        header='import tonic_amd as amd, tonic_amd.torch, stub_agents; amd.install()',
        agent='stub_agents.LoggingOnlyPPO(update_every=8, iterations=3)',
        environment='__import__("tonic_amd").environments.Synthetic(5, 2, max_episode_steps=7)',
        test_environment=None,
        trainer='tonic.Trainer(steps=48, epoch_steps=48, save_steps=48, show_progress=False)',
        before_training=None, after_training=None, parallel=2, sequential=3, seed=3,
        name='run', environment_name='env', checkpoint='last', path=None)
    assert seen and all(seen), 'the agent must be handed views of the shared block'
    assert (tmp_path / 'env' / 'run' / '3' / 'log.csv').exists()


@pytest.mark.parametrize('name', ['PPO', 'A2C', 'TRPO', 'SAC', 'TD3', 'DDPG', 'D4PG', 'MPO'])
def test_reference_agents_load_this_packages_checkpoints(tmp_path, name):
    """tonic/torch/agents/agent.py:23-26: the reference agent's strict load_state_dict accepts a
    `.pt` written by this package's Agent.save (same keys, same shapes)."""
    import torch
    tonic = reference_loader.load_reference()
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    observation_space, action_space = Box(-np.inf, np.inf, (11,)), Box(-1, 1, (3,))
    mine = getattr(tonic_amd.torch.agents, name)()
    mine.model.initialize(observation_space, action_space)         # CPU parameters suffice
    torch.manual_seed(1)
    for parameter in mine.model.parameters():
        parameter.data.copy_(torch.randn(parameter.shape))
    mine.save(str(tmp_path / 'step_1'))
    theirs = getattr(tonic.torch.agents, name)()
    theirs.initialize(observation_space, action_space, seed=0)
    theirs.load(str(tmp_path / 'step_1'))                           # strict: keys and shapes
    want = mine.model.state_dict()
    got = theirs.model.state_dict()
    assert set(got) == set(want)
    for key in want:
        assert torch.equal(got[key], want[key].cpu()), key


def test_categorical_with_support_equals_the_reference():
    """models.CategoricalWithSupport / DistributionalValueHead (critics.py:23-66): same support,
    probabilities, mean and projection as the reference's, bit for bit, on returns inside, outside
    and exactly on the support."""
    import torch
    tonic = reference_loader.load_reference()
    import tonic_amd.torch as tt
    torch.manual_seed(3)
    theirs = tonic.torch.models.DistributionalValueHead(-7.5, 4.0, 23)
    mine = tt.models.DistributionalValueHead(-7.5, 4.0, 23)
    assert torch.equal(theirs.values, mine.values)
    logits = torch.randn(40, 23) * 2
    a = tonic.torch.models.critics.CategoricalWithSupport(theirs.values, logits)
    b = tt.models.CategoricalWithSupport(mine.values, logits)
    returns = torch.randn(40, 23) * 6
    returns[0] = theirs.values                       # exactly on the atoms
    returns[1] = -100.0
    returns[2] = 100.0
    assert torch.equal(a.probabilities, b.probabilities)
    assert torch.equal(a.mean(), b.mean())
    assert torch.equal(a.project(returns), b.project(returns))


def test_explorations_and_scripted_agents_follow_the_reference_streams():
    """tonic/explorations/noisy.py and tonic/agents/basic.py (host logic either side of the policy
    forward; `--agent 'tonic.agents.UniformRandom()'` is part of the reference's command line): same
    constructor arguments, same NumPy RandomState stream, same per-worker reset
    handling — bit-identical action sequences over warm-up, policy steps and resets."""
    tonic = reference_loader.load_reference()
    import tonic_amd
    from tonic_amd.environments import Box
    action_space = Box(-1, 1, (3,))
    policy = lambda observations: np.tanh(observations[:, :3] * 0.7).astype(np.float32)
    rng = np.random.RandomState(5)
    for name, kwargs in (('NoActionNoise', dict(start_steps=3)),
                         ('NormalActionNoise', dict(scale=0.3, start_steps=3)),
                         ('OrnsteinUhlenbeckActionNoise', dict(scale=0.4, clip=1.5, start_steps=3))):
        theirs = getattr(tonic.explorations, name)(**kwargs)
        ours = getattr(tonic_amd.explorations, name)(**kwargs)
        theirs.initialize(policy, action_space, seed=11)
        ours.initialize(policy, action_space, seed=11)
        for steps in range(12):
            observations = rng.standard_normal((4, 5)).astype(np.float32)
            a, b = theirs(observations, steps), ours(observations, steps)
            assert a.dtype == b.dtype and np.array_equal(a, b), (name, steps)
            resets = rng.rand(4) < 0.3
            theirs.update(resets)
            ours.update(resets)
    for name, kwargs in (('NormalRandom', dict(loc=0.1, scale=0.5)), ('UniformRandom', {}),
                         ('OrnsteinUhlenbeck', dict(scale=0.3)), ('Constant', dict(constant=0.25))):
        theirs = getattr(tonic.agents, name)(**kwargs)
        ours = getattr(tonic_amd.agents, name)(**kwargs)
        theirs.initialize(None, action_space, seed=3)
        ours.initialize(None, action_space, seed=3)
        for steps in range(8):
            observations = np.zeros((4, 5), np.float32)
            assert np.array_equal(theirs.step(observations, steps), ours.step(observations, steps))
            assert np.array_equal(theirs.test_step(observations, steps),
                                  ours.test_step(observations, steps))
            resets = rng.rand(4) < 0.3
            outcome = (observations, np.zeros(4), resets, resets, steps)
            theirs.update(*outcome), ours.update(*outcome)
            theirs.test_update(*outcome), ours.test_update(*outcome)


def test_learning_curve_fixture_is_what_the_reference_does(tmp_path):
    """tests/golden/learning_curves.json (what tests/test_gpu_learning.py holds this package's agents
    against) is regenerated for one cheap case by training the unmodified reference here, and must
    come out identical: the fixture is pinned on the reference, not on a past run of this package."""
    import json
    import reach_task
    tonic = reference_loader.load_reference()
    import tonic.torch
    import torch
    torch.set_num_threads(8)                    # as oracle/make_learning_curves.py
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                                         'learning_curves.json')))
    assert set(golden['curves']) == set(reach_task.CASES)
    agent = reach_task.build_agent(tonic, tonic.torch.agents, 'A2C')
    curve = reach_task.train(tonic, agent, 'A2C', str(tmp_path))
    np.testing.assert_allclose(curve, golden['curves']['A2C'], rtol=0, atol=1e-6)


def test_committed_goldens_are_what_their_generator_writes(tmp_path):
    """oracle/make_golden.py run NOW on the unmodified reference (in a subprocess: one torch
    thread, its own generator state) writes exactly tests/golden/*.npz — same keys, same bits.
    A generator that gained a field (or a reference / torch build that computes something else)
    fails here instead of leaving the committed fixtures silently behind."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, %r); import make_golden; make_golden.OUT = %r; '
            'sys.argv = sys.argv[:1]; make_golden.main()'
            % (os.path.join(root, 'oracle'), str(tmp_path)))
    subprocess.run([sys.executable, '-c', code], check=True, capture_output=True, timeout=600)
    committed = os.path.join(root, 'tests', 'golden')
    fresh = sorted(f for f in os.listdir(tmp_path) if f.endswith('.npz'))
    assert fresh == sorted(f for f in os.listdir(committed) if f.endswith('.npz'))
    for name in fresh:
        got = np.load(os.path.join(tmp_path, name), allow_pickle=True)
        want = np.load(os.path.join(committed, name), allow_pickle=True)
        assert sorted(got.files) == sorted(want.files), name
        for key in got.files:
            a, b = got[key], want[key]
            same = (np.array_equal(a, b, equal_nan=True) if a.dtype.kind in 'fc'
                    else np.array_equal(a, b))
            assert same, (name, key)
