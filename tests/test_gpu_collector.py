"""GPU tests of the pinned-host collector (tonic_collector_* through ctypes): the host-in-the-loop
path — shared block, fused act + deferred outcome store, completion word / event instead of a
stream sync — against the CPU oracle and against the device-resident collect entry points."""
import ctypes

import numpy as np
import pytest

import numpy_port as port

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from tonic_amd import _lib
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return _lib.load()


def _in_a_process_of_its_own(request):
    """Runs the calling test in a fresh interpreter (pytest on its node id) and returns True when it
    did — the outer call then has nothing left to do.  For tests whose POINT is that two kernels run
    side by side: HIP maps the streams of a process onto a handful of hardware queues, a queue runs
    its kernels one after the other, and in a process that has created dozens of streams (this file's
    other tests) a foreign kernel and the collector's resident kernel may share one — then there is
    no concurrency to test."""
    import os
    import subprocess
    import sys
    if os.environ.get('TONIC_AMD_TEST_INNER') == '1':
        return False
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run(
        [sys.executable, '-m', 'pytest', '-q', '-x', '-s', '-p', 'no:cacheprovider', '-m', 'gpu',
         request.node.nodeid], cwd=root, env=dict(os.environ, TONIC_AMD_TEST_INNER='1'),
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(done.stdout[-1500:])
    assert done.returncode == 0, done.stdout[-4000:]
    return True


def _actor(O, A, seed):
    rng = np.random.RandomState(seed)
    shapes = [(64, O), (64,), (64, 64), (64,), (1, A), (A, 64), (A,)]
    return [(rng.standard_normal(s) * (0.3 if len(s) == 2 else 0.1)).astype(np.float32)
            for s in shapes]


def _segment(T, W, O, A):
    def new(*shape):
        return torch.full(shape, np.nan, dtype=torch.float32, device='cuda')
    return dict(observations=new(T, W, O), actions=new(T, W, A), next_observations=new(T, W, O),
                rewards=new(T, W), resets=new(T, W), terminations=new(T, W), log_probs=new(T, W))


WIDE_STEPS = [pytest.param(0, O, A, W, id=f'wide-{O}-{A}-{W}')
              for O, A, W in ((111, 8, 256), (376, 17, 67), (40, 21, 5), (33, 3, 300))]


@pytest.mark.parametrize('transport,O,A,W',
                         [(t, O, A, W) for t in (0, 1, 2, 3)
                          for O, A, W in ((17, 6, 256), (28, 8, 1280), (3, 1, 5))] + WIDE_STEPS)
def test_collector_steps_match_the_oracle(lib, transport, O, A, W):
    """T host-in-the-loop steps through the C entry points: actions and log-probs against
    numpy_port.ppo_act (the reference's forward + sample + log_prob), stored rows and outcome
    rows bit-exact copies, normaliser sums bit-exact against MeanStdPort.record.  The wide cases
    (O > 32 or A > 8) take the collector's layer-by-layer steps (ingest, dense x 3, sample + store)."""
    from tonic_amd import _lib
    from tonic_amd.collector import Block, Collector
    T = 5
    rng = np.random.RandomState(O * 1000 + W)
    params = _actor(O, A, 7)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in params])).cuda()
    block = Block(W, O, A)
    collector = Collector(block, transport)
    seg = _segment(T, W, O, A)
    sums = torch.zeros(2 * O, device='cuda')
    collector.bind_segment(seg, sums, T)
    torch.cuda.synchronize()
    collector.begin_rollout(flat)
    want = {k: np.full(tuple(v.shape), np.nan, np.float32) for k, v in seg.items()}
    recorder = port.MeanStdPort((O,))
    for t in range(T):
        observations = rng.standard_normal((W, O)).astype(np.float32)
        eps = rng.standard_normal((W, A)).astype(np.float32)
        block.observations[:] = observations
        block.eps[t & 1][:] = eps
        collector.ppo_step(t, t & 1, t > 0)
        collector.wait_actions()
        actions = block.actions.copy()
        want_actions, want_log_probs = port.ppo_act(params, observations, eps)
        # (wide cases: hundreds of float32 products per pre-activation, weights of 0.3 each)
        np.testing.assert_allclose(actions, want_actions, rtol=0, atol=3e-6 if O <= 32 else 5e-5)
        want['observations'][t] = observations
        want['actions'][t] = actions
        want['log_probs'][t] = want_log_probs
        recorder.record(observations)
        # the environment's answer to these actions: the outcome of step t
        outcome = dict(next_observations=rng.standard_normal((W, O)).astype(np.float32),
                       rewards=rng.standard_normal(W).astype(np.float32),
                       resets=(rng.uniform(size=W) < 0.3).astype(np.float32),
                       terminations=(rng.uniform(size=W) < 0.1).astype(np.float32))
        for key, value in outcome.items():
            getattr(block, key)[:] = value
            want[key][t] = value
    collector.end_rollout(T - 1)
    torch.cuda.synchronize()
    for key in ('observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations'):
        assert np.array_equal(seg[key].cpu().numpy(), want[key]), key
    np.testing.assert_allclose(seg['log_probs'].cpu().numpy(), want['log_probs'], rtol=0,
                               atol=2e-5 if O <= 32 else 2e-4)
    got = sums.cpu().numpy()
    assert np.array_equal(got[:O], recorder.new_sum) and np.array_equal(got[O:], recorder.new_sum_sq)
    collector.close()


@pytest.mark.parametrize('transport,O,A,W', [(t, O, A, W) for t in (0, 2, 3)
                                             for O, A, W in ((28, 8, 1280), (17, 6, 1024), (3, 2, 6000))])
def test_carried_over_rows_cross_pcie_once(lib, transport, O, A, W):
    """Many workers + a block whose writer promised carry-over (tonic_collector_block_carry_over):
    the act launch stores the observation rows of the workers that did NOT reset as the previous
    step's next observations, the copy role fetches the rows of those that did.  The Segment must
    hold bit-exact copies of what an environment following the reference's protocol
    (distributed.py:41-57) wrote — including rows that reset and a last row stored by end_rollout."""
    from tonic_amd.collector import Block, Collector
    T = 6
    rng = np.random.RandomState(O * 1000 + W)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in _actor(O, A, 7)])).cuda()
    block = Block(W, O, A)
    block.promise_carry_over()
    collector = Collector(block, transport)
    seg = _segment(T, W, O, A)
    sums = torch.zeros(2 * O, device='cuda')
    collector.bind_segment(seg, sums, T)
    torch.cuda.synchronize()
    collector.begin_rollout(flat)
    want = {k: np.full(tuple(v.shape), np.nan, np.float32) for k, v in seg.items()}
    observations = rng.standard_normal((W, O)).astype(np.float32)
    for t in range(T):
        block.observations[:] = observations
        block.eps[t & 1][:] = rng.standard_normal((W, A)).astype(np.float32)
        collector.ppo_step(t, t & 1, t > 0)
        collector.wait_actions()
        want['observations'][t] = observations
        # the environment's answer: next observations, flags; reset workers start over
        next_observations = rng.standard_normal((W, O)).astype(np.float32)
        resets = rng.uniform(size=W) < (0.0 if t == 2 else 0.3 if t != 3 else 1.0)
        outcome = dict(next_observations=next_observations,
                       rewards=rng.standard_normal(W).astype(np.float32),
                       resets=resets.astype(np.float32),
                       terminations=(resets & (rng.uniform(size=W) < 0.5)).astype(np.float32))
        for key, value in outcome.items():
            getattr(block, key)[:] = value
            want[key][t] = value
        observations = next_observations.copy()
        observations[resets] = rng.standard_normal((int(resets.sum()), O)).astype(np.float32)
    collector.end_rollout(T - 1)
    torch.cuda.synchronize()
    for key in ('observations', 'next_observations', 'rewards', 'resets', 'terminations'):
        assert np.array_equal(seg[key].cpu().numpy(), want[key]), key
    collector.close()


def test_collector_equals_device_resident_collect(lib):
    """The host-in-the-loop launch is the device-resident packed collect kernel with the outcome
    deferred by one step: identical Segment bits for identical inputs, both transports."""
    from tonic_amd import _lib
    from tonic_amd.collector import Block, Collector
    O, A, W, T = 17, 6, 256, 6
    rng = np.random.RandomState(3)
    params = _actor(O, A, 11)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in params])).cuda()
    obs = rng.standard_normal((T + 1, W, O)).astype(np.float32)
    eps = rng.standard_normal((T, W, A)).astype(np.float32)
    rewards = rng.standard_normal((T, W)).astype(np.float32)
    resets = (rng.uniform(size=(T, W)) < 0.2).astype(np.float32)
    terms = resets * (rng.uniform(size=(T, W)) < 0.5)
    p = _lib.ptr
    ref = _segment(T, W, O, A)
    ref_sums = torch.zeros(2 * O, device='cuda')
    packed = torch.empty(lib.tonic_ppo_packed_actor_floats(O, A), device='cuda')
    _lib.check(lib.tonic_ppo_pack_actor(p(flat), p(packed), O, A, None), 'pack')
    d = [torch.as_tensor(x).cuda() for x in (obs, eps, rewards, resets, terms.astype(np.float32))]
    _lib.check(lib.tonic_ppo_collect_steps_packed(
        p(packed), p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(ref['observations']),
        p(ref['actions']), p(ref['next_observations']), p(ref['rewards']), p(ref['resets']),
        p(ref['terminations']), p(ref['log_probs']), p(ref_sums), 0, T, W, O, A, None), 'collect')
    torch.cuda.synchronize()
    for transport in (0, 1, 2, 3):
        block = Block(W, O, A)
        collector = Collector(block, transport)
        seg = _segment(T, W, O, A)
        sums = torch.zeros(2 * O, device='cuda')
        collector.bind_segment(seg, sums, T)
        torch.cuda.synchronize()
        collector.begin_rollout(flat)
        for t in range(T):
            block.observations[:] = obs[t]
            block.eps[t & 1][:] = eps[t]
            collector.ppo_step(t, t & 1, t > 0)
            collector.wait_actions()
            block.next_observations[:] = obs[t + 1]
            block.rewards[:] = rewards[t]
            block.resets[:] = resets[t]
            block.terminations[:] = terms[t]
        collector.end_rollout(T - 1)
        torch.cuda.synchronize()
        for key in seg:
            assert np.array_equal(seg[key].cpu().numpy(), ref[key].cpu().numpy()), (transport, key)
        assert np.array_equal(sums.cpu().numpy(), ref_sums.cpu().numpy())
        assert np.array_equal(block.actions, ref['actions'][T - 1].cpu().numpy())
        collector.close()


def test_parallel_workers_feed_the_gpu_in_place(lib):
    """Forked worker groups write into the shared block, the agent page-locks that very block
    (identity-bound views, no host copy) and the stored Segment equals the one collected through
    the Sequential collector with the same seeds."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments
    O, A, T = 17, 6, 12
    segments = []
    for groups, per_group in ((3, 2), (1, 6)):
        env = environments.distribute(
            lambda: environments.Synthetic(O, A, max_episode_steps=5), groups, per_group)
        env.initialize(seed=4)
        agent = tonic_amd.torch.agents.PPO(
            replay=tonic_amd.replays.Segment(size=T + 1, batch_iterations=2))
        agent.initialize(env.observation_space, env.action_space, seed=9)
        observations = env.start()
        for t in range(T):
            actions = agent.step(observations, t * 6)
            assert agent._block is env.block, 'the agent must adopt the environment block'
            observations, infos = env.step(actions)
            agent.update(**infos, steps=t * 6)
        agent._settle()                 # (update() has issued step T already: let it finish)
        agent._collector.end_rollout(T - 1)
        torch.cuda.synchronize()
        segments.append({k: agent.replay.buffers[k][:T].cpu().numpy() for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
            'log_probs')})
        sums = agent.model.observation_normalizer.device_sums.cpu().numpy()
        segments[-1]['sums'] = sums
        if groups > 1:
            env.close()
    for key, want in segments[1].items():
        assert np.array_equal(segments[0][key], want), key
    assert segments[0]['resets'].sum() > 0


def test_test_step_keeps_the_reference_noise_order(lib):
    """The next step's noise is drawn ahead; a test episode in between must see the generator
    where the reference would have it (a2c.py:87-90 draws from the same stream)."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    O, A, W = 5, 2, 8
    rng = np.random.RandomState(0)
    obs = rng.standard_normal((4, W, O)).astype(np.float32)
    test_obs = rng.standard_normal((1, O)).astype(np.float32)

    def run(with_test):
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=64))
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
        out = []
        for t in range(3):
            out.append(agent.step(obs[t], t * W))
            agent.update(obs[t + 1], np.zeros(W, np.float32), np.zeros(W, bool),
                         np.zeros(W, bool), steps=t * W)
            if with_test and t == 1:
                out.append(agent.test_step(test_obs, t * W))
        return out

    got = run(True)
    plain = run(False)
    # steps 0 and 1 agree; the test draw sits between step 1 and step 2 in the stream
    assert np.array_equal(got[0], plain[0]) and np.array_equal(got[1], plain[1])
    assert not np.array_equal(got[3], plain[2])
    agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=64))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
    state = agent.model.state_dict()
    params = [state[k].detach().cpu().numpy() for k in (
        'actor.torso.model.0.weight', 'actor.torso.model.0.bias', 'actor.torso.model.2.weight',
        'actor.torso.model.2.bias', 'actor.head.log_scale', 'actor.head.loc_layer.0.weight',
        'actor.head.loc_layer.0.bias')]
    # agent.initialize seeds the generator, then the model initialisation consumes draws: replay
    # the expected noise from the generator state right after initialisation instead
    after_init = torch.get_rng_state()
    eps = [torch.randn(W, A).numpy() for _ in range(2)]
    eps_test = torch.randn(1, A).numpy()
    eps_last = torch.randn(W, A).numpy()
    torch.set_rng_state(after_init)
    want0, _ = port.ppo_act(params, obs[0], eps[0])
    want1, _ = port.ppo_act(params, obs[1], eps[1])
    want_test, _ = port.ppo_act(params, test_obs, eps_test)
    want2, _ = port.ppo_act(params, obs[2], eps_last)
    for got_actions, want in zip(got, (want0, want1, want_test, want2)):
        np.testing.assert_allclose(got_actions, want, rtol=0, atol=3e-6)


def test_a_closed_agent_lets_go_and_leaves_the_generator_where_the_reference_has_it(lib):
    """agent.close(): the noise helper thread ends and no longer keeps the agent (its Segment, its
    collector, the page-locked block) alive — bench.py's second job on a shared device crawled behind
    the first one's leftovers — the generator is rewound to the draws the agent has consumed (the
    helper runs up to 128 steps ahead), and the agent works again after a new bind."""
    import gc
    import threading
    import weakref
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    O, A, W = 5, 2, 8                                   # W * A = 16: the block-draw mode
    rng = np.random.RandomState(1)
    obs = rng.standard_normal((5, W, O)).astype(np.float32)

    def helpers():
        return [t for t in threading.enumerate() if t.name == 'tonic-noise-ahead' and t.is_alive()]

    before = len(helpers())
    agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=64))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
    after_init = torch.get_rng_state()
    first = []
    for t in range(3):
        first.append(agent.step(obs[t], t * W).copy())
        agent.update(obs[t + 1], np.zeros(W, np.float32), np.zeros(W, bool), np.zeros(W, bool),
                     steps=t * W)
    assert len(helpers()) == before + 1
    agent.close()
    consumed = torch.get_rng_state()
    torch.set_rng_state(after_init)
    for _ in range(3):                                  # the three steps' draws, nothing more
        torch.randn(W, A)
    assert torch.equal(torch.get_rng_state(), consumed)
    # ... the agent binds again and goes on with the stream
    again = agent.step(obs[3], 3 * W)
    assert np.isfinite(again).all() and again.shape == (W, A)
    agent.close()
    ref = weakref.ref(agent)
    del agent
    gc.collect()
    assert ref() is None, 'a closed agent is still referenced (helper thread?)'
    for thread in helpers():
        thread.join(timeout=2.0)
    assert len(helpers()) == before


def test_step_issued_from_update_is_bit_identical(lib, monkeypatch):
    """With a block-backed environment `agent.update` issues the next step's launch itself
    (the observations are already in the block, the noise is drawn ahead) and `agent.step` only
    collects the result — or issues the step again when a test episode came in between or other
    observations are handed over.  Everything a rollout leaves behind must be bit-identical to
    the run with TONIC_AMD_SPECULATE=0: actions, Segment rows, normaliser sums, learner update."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments
    O, A, W, T = 6, 3, 12, 10

    def run(speculate):
        monkeypatch.setenv('TONIC_AMD_SPECULATE', '1' if speculate else '0')
        env = environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=4), 1, W)
        env.initialize(seed=3)
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=2))
        agent.initialize(env.observation_space, env.action_space, seed=9)
        observations = env.start()
        rng = np.random.RandomState(1)
        test_obs = rng.standard_normal((2, O)).astype(np.float32)
        trace, issued_early = [], 0
        for t in range(2 * T + 3):
            if t == 5:                                   # foreign observations: a copy of the block's
                observations = observations.copy()
            actions = agent.step(observations, t * W)
            trace.append(actions.copy())
            observations, infos = env.step(actions)
            agent.update(**infos, steps=t * W)
            issued_early += bool(agent._speculated)
            if t in (2, 7, 13):                          # test episodes between update and step
                trace.append(agent.test_step(test_obs, t * W))
            if t == T - 2:
                kept = {k: v.clone() for k, v in agent.replay.buffers.items()}
        torch.cuda.synchronize()
        state = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
        return trace, kept, state, issued_early

    with_early, kept_a, state_a, early = run(True)
    plain, kept_b, state_b, none = run(False)
    assert early >= T and none == 0, 'the early launch must actually be exercised'
    assert len(with_early) == len(plain)
    for a, b in zip(with_early, plain):
        assert np.array_equal(a, b)
    for key in ('observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
                'log_probs'):
        assert torch.equal(kept_a[key][:T - 2], kept_b[key][:T - 2]), key
    for key in state_a:                                  # two learner updates incl. the normaliser
        assert torch.equal(state_a[key], state_b[key]), key


@pytest.mark.parametrize('kind', ['sequential', 'parallel', 'batch'])
def test_step_issued_by_the_environment_is_bit_identical(lib, monkeypatch, kind):
    """In the steady state of a block-fed loop `agent.step` leaves the NEXT step's command in the
    block's header and the environment issues it the moment its step record is complete
    (Sequential / SyntheticBatch: at the end of their step; Parallel: the last worker group, before
    the parent wakes up); `agent.update` only confirms.  Everything a rollout leaves behind must be
    bit-identical to the run in which update() issues the command itself (TONIC_AMD_ARM=0) — with
    test episodes, foreign observations and episode ends in between."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments
    O, A, W, T = 6, 3, 12, 10

    def build():
        if kind == 'batch':
            return environments.SyntheticBatch(W, O, A, max_episode_steps=7, pool=5)
        groups = 2 if kind == 'parallel' else 1
        return environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=4),
                                       groups, W // groups)

    def run(arm):
        monkeypatch.setenv('TONIC_AMD_ARM', '1' if arm else '0')
        env = build()
        env.initialize(seed=3)
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=2))
        agent.initialize(env.observation_space, env.action_space, seed=9)
        observations = env.start()
        rng = np.random.RandomState(1)
        test_obs = rng.standard_normal((2, O)).astype(np.float32)
        trace = []
        for t in range(2 * T + 3):
            if t == 5:                                   # foreign observations: a copy of the block's
                observations = observations.copy()
            actions = agent.step(observations, t * W)
            trace.append(actions.copy())
            observations, infos = env.step(actions)
            if t == 15:                                  # foreign infos after the environment rang
                infos = {k: v.copy() for k, v in infos.items()}
            if t == 17:                                  # a test episode between step and update
                trace.append(agent.test_step(test_obs, t * W))
            agent.update(**infos, steps=t * W)
            if t in (2, 7, 13):                          # test episodes between update and step
                trace.append(agent.test_step(test_obs, t * W))
            if t == T - 2:
                kept = {k: v.clone() for k, v in agent.replay.buffers.items()}
        torch.cuda.synchronize()
        state = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
        rung = agent.steps_issued_by_environment
        agent.close()
        if hasattr(env, 'close'):
            env.close()
        return trace, kept, state, rung

    armed, kept_a, state_a, rung = run(True)
    plain, kept_b, state_b, none = run(False)
    assert rung >= 6 and none == 0, 'the environment must actually have issued steps'
    assert len(armed) == len(plain)
    for a, b in zip(armed, plain):
        assert np.array_equal(a, b)
    for key in ('observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
                'log_probs'):
        assert torch.equal(kept_a[key][:T - 2], kept_b[key][:T - 2]), key
    for key in state_a:                                  # two learner updates incl. the normaliser
        assert torch.equal(state_a[key], state_b[key]), key


@pytest.mark.parametrize('O,A,W,T', [(17, 6, 32, 24), (28, 8, 1280, 40)])
def test_critic_iterations_under_the_next_rollout_are_bit_identical(lib, monkeypatch, O, A, W, T):
    """On one GPU with full-batch iterations `PPO.update` runs the actor's iterations, leaves the
    critic's to a second stream behind them and returns; they run while the next rollout is
    collected (into a spare observation buffer, with a snapshot of the normaliser).  Three rollouts
    + updates must leave exactly what the interleaved launches leave (TONIC_AMD_CRITIC_OVERLAP=0):
    actions, parameters incl. the normaliser, and every logged row — the critic's rows land in the
    logger before anything reads them (settle(): `last_infos`, the next update, the logger's dump)."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments

    def run(overlap):
        monkeypatch.setenv('TONIC_AMD_CRITIC_OVERLAP', '1' if overlap else '0')
        env = environments.SyntheticBatch(W, O, A, max_episode_steps=10, pool=7)
        env.initialize(seed=3)
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=6))
        agent.initialize(env.observation_space, env.action_space, seed=9)
        observations = env.start()
        trace, rows, overlapped = [], [], 0
        for t in range(3 * T + 5):
            actions = agent.step(observations, t * W)
            trace.append(actions.copy())
            observations, infos = env.step(actions)
            agent.update(**infos, steps=t * W)
            if (t + 1) % T == 0:
                overlapped += getattr(agent, '_critic_pending', None) is not None
                if (t + 1) // T == 2:
                    rows.append(np.array(agent.last_infos))          # settles
        torch.cuda.synchronize()
        agent.settle()
        rows.append(np.array(agent.last_infos))
        state = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
        agent.close()
        return trace, rows, state, overlapped

    with_overlap, rows_a, state_a, overlapped = run(True)
    plain, rows_b, state_b, none = run(False)
    assert overlapped == 3 and none == 0, 'the overlap must actually be exercised'
    for a, b in zip(with_overlap, plain):
        assert np.array_equal(a, b)
    for a, b in zip(rows_a, rows_b):
        assert np.array_equal(a, b)
    for key in state_a:
        assert torch.equal(state_a[key], state_b[key]), key


def test_critic_chain_under_a_running_rollout_is_bit_identical_at_size(lib, monkeypatch, request):
    """The mode bench.py measures, at a size where the critic's chain REALLY runs under the next
    rollout: an update of 262 144 transitions x 80 iterations leaves ~6 ms of critic launches on the
    second stream while the host drives the next rollout (whose first steps also take 0.3 ms of
    simulator time each, so that the resident collect kernel parks and is launched again under the
    chain).  The critic's launches have the same width in both modes (`PPO._critic_width`: 219 of
    256 workgroups at 256 workers), so the comparison with the interleaved launches
    (TONIC_AMD_CRITIC_OVERLAP=0) is BIT FOR BIT — every logged row of the first and the last update
    (80 actor rows, 80 critic rows each), every parameter and the normaliser after three rollouts +
    updates.  (From the second update on a gate holds the chain back so that it ends a margin before
    the rollout does: PPO._arm_gate.)  A race
    on the spare observation buffer, the normaliser snapshot or the Segment cannot hide behind a
    tolerance here."""
    if _in_a_process_of_its_own(request):
        return
    import time
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments
    O, A, W, T = 17, 6, 256, 1024

    def run(overlap):
        import gc
        gc.collect()                    # (agents of earlier tests: their close() synchronises the device)
        torch.cuda.synchronize()
        monkeypatch.setenv('TONIC_AMD_CRITIC_OVERLAP', '1' if overlap else '0')
        env = environments.SyntheticBatch(W, O, A, max_episode_steps=1000, pool=5)
        env.initialize(seed=3)
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=80))
        agent.initialize(env.observation_space, env.action_space, seed=9)
        observations = env.start()
        in_flight, widths, first = 0, [], None
        for t in range(3 * T + 40):
            actions = agent.step(observations, t * W)
            observations, infos = env.step(actions)
            if t >= T and t % 16 == 0:                          # somewhere under a rollout
                pending = getattr(agent, '_critic_pending', None)
                in_flight += (pending is not None and pending['done'] is not None
                              and not pending['done'].query())
            if T <= t < T + 30 or 2 * T <= t < 2 * T + 30:      # right behind an update
                time.sleep(0.0003)
            agent.update(**infos, steps=t * W)
            if t == T - 1:
                widths.append(agent.critic_updater.max_workgroups)
            if t == 2 * T - 2:                                   # the end of the second rollout
                first = np.array(agent.last_infos)              # (the first update's rows; settled long ago)
        torch.cuda.synchronize()
        rows = np.array(agent.last_infos)
        state = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
        assert (getattr(agent, '_critic_stream', None) is not None) == overlap
        widths.append(agent.critic_updater.max_workgroups)
        agent.close()
        return first, rows, state, in_flight, widths

    first_b, rows_b, state_b, none, widths_b = run(False)
    assert none == 0 and widths_b == [256 - (W // 16 + 5) - 16] * 2
    seen = []
    for attempt in range(3):
        # EVERY overlapped run must give the interleaved run's bits; at least one of (at most) three
        # must have had the host's polls find the chain still running under the rollout's steps
        # (whether a poll falls inside the chain's ~6 ms depends on what else this process is doing:
        # informative per run, required once)
        first_a, rows_a, state_a, in_flight, widths_a = run(True)
        seen.append(in_flight)
        assert widths_a == widths_b
        assert np.isfinite(rows_a).all() and (rows_a[1, :, 0] > 0).all()
        # the first update: ten iterations a reader can check by eye (critic loss / v), then all rows
        assert np.array_equal(first_a[1, :10, :2], first_b[1, :10, :2])
        assert np.array_equal(first_a, first_b)
        assert np.array_equal(rows_a, rows_b)
        for key in state_a:
            assert torch.equal(state_a[key], state_b[key]), key
        if in_flight >= 1:
            break
    print('sampled steps that found the critic chain running:', seen)
    assert max(seen) >= 1, 'the chain must actually have run under a rollout'


def test_stream_gate_holds_a_stream_until_the_host_stores(lib):
    """tonic_stream_gate: work behind the gate starts when the host stores the ticket into the pinned
    word — a plain CPU store — or when the gate's own time is up, and not before."""
    import time
    from tonic_amd import _lib
    word = torch.zeros(1, dtype=torch.int32).pin_memory()
    view = word.numpy()
    side = torch.cuda.Stream()
    target = torch.zeros(1024, device='cuda')
    torch.cuda.synchronize()
    for ticket, opened_by_host in ((1, True), (2, False), (3, True)):
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            _lib.check(lib.tonic_stream_gate(word.data_ptr(), ticket, 0.25, side.cuda_stream), 'gate')
            target.add_(1.0)
            done.record(side)
        time.sleep(0.05)
        assert not done.query(), 'the gate must hold the stream'
        t0 = time.perf_counter()
        if opened_by_host:
            view[0] = ticket
        done.synchronize()
        waited = time.perf_counter() - t0
        assert (waited < 0.05) if opened_by_host else (0.1 < waited < 0.4), (ticket, waited)
    assert float(target[0]) == 3.0
    # a ticket that is already there does not hold anything
    with torch.cuda.stream(side):
        _lib.check(lib.tonic_stream_gate(word.data_ptr(), 3, 5.0, side.cuda_stream), 'gate')
    t0 = time.perf_counter()
    side.synchronize()
    assert time.perf_counter() - t0 < 0.05


def test_completion_words_order_the_actions(lib):
    """The host must never read actions older than the completion words it waited for: many
    steps, host copy of the block's actions / rewards against what the kernels stored."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts',
                        'collector_stress.py')
    spec = importlib.util.spec_from_file_location('collector_stress', path)
    stress = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stress)
    assert stress.main(W=6, steps=6144) == 0
    assert stress.main(W=256, steps=2048) == 0
    assert stress.main(W=256, steps=1024, transport=1) == 0
    assert stress.main(W=6, steps=6144, transport=2) == 0
    assert stress.main(W=256, steps=4096, transport=2) == 0
    assert stress.main(W=6, steps=6144, transport=3) == 0
    assert stress.main(W=256, steps=4096, transport=3) == 0
    assert stress.main(W=256, O=28, A=8, steps=4096, transport=3) == 0


def test_resident_kernel_parks_and_resumes(lib):
    """transport 2: the resident collect kernel leaves after ~200 us without a command (a slow
    simulator, a test episode, the end of training) and the next step starts it again; a rollout
    may also end while it is parked.  Same Segment as the launch-per-step transport."""
    import time
    from tonic_amd.collector import Block, Collector
    O, A, W, T = 17, 6, 48, 9
    rng = np.random.RandomState(1)
    params = _actor(O, A, 3)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in params])).cuda()
    obs = rng.standard_normal((T + 1, W, O)).astype(np.float32)
    eps = rng.standard_normal((T, W, A)).astype(np.float32)
    rewards = rng.standard_normal((T, W)).astype(np.float32)
    results = {}
    for transport, pauses in ((0, ()), (2, (2, 3, 6, 8)), (3, (2, 3, 6, 8))):
        block = Block(W, O, A)
        collector = Collector(block, transport)
        seg = _segment(T, W, O, A)
        sums = torch.zeros(2 * O, device='cuda')
        collector.bind_segment(seg, sums, T)
        torch.cuda.synchronize()
        collector.begin_rollout(flat)
        for t in range(T):
            if t in pauses:
                time.sleep(0.02)                  # 100 x the park time
            block.observations[:] = obs[t]
            block.eps[t & 1][:] = eps[t]
            collector.ppo_step(t, t & 1, t > 0)
            collector.wait_actions()
            block.next_observations[:] = obs[t + 1]
            block.rewards[:] = rewards[t]
            block.resets[:] = 0
            block.terminations[:] = 0
        if pauses:
            time.sleep(0.02)                      # the rollout ends while the kernel is parked
        collector.end_rollout(T - 1)
        torch.cuda.synchronize()
        results[transport] = {k: v.cpu().numpy() for k, v in seg.items()}
        results[transport]['sums'] = sums.cpu().numpy()
        collector.close()
    for key, want in results[0].items():
        assert np.array_equal(results[2][key], want), key
        assert np.array_equal(results[3][key], want), key


@pytest.mark.parametrize('O,A,W,T,hog,hog_ms', [
    (28, 8, 1280, 10, 200, 80.0),     # 85 slots, 56 compute units: the whole rollout beside the foreign kernel
    (28, 8, 1280, 160, 200, 4.0),     # ... which leaves in the middle: the late workgroups join
    (17, 6, 256, 300, 245, 3.0),      # 21 slots, 11 compute units, then everybody
    (3, 1, 5, 40, 252, 5.0)])         # 6 slots, 4 compute units
def test_resident_kernel_runs_the_slots_of_absent_workgroups(lib, request, O, A, W, T, hog, hog_ms):
    """The resident collect kernel next to a FOREIGN kernel that holds most of the chip before the
    first command (tonic_debug_occupy: `hog` workgroups of 100 KB LDS, one per compute unit): only
    some of the launch's workgroups find a compute unit, the others get in when the foreign kernel
    ends — or never during the rollout.  Every step still completes (the workgroups that are there
    run the slots of those that are not, claim by claim; a late workgroup joins and skips what was
    run for it) and the Segment, the normaliser sums and the actions are those of the undisturbed
    launch-per-step transport, bit for bit."""
    if _in_a_process_of_its_own(request):
        return
    import time
    from tonic_amd import _lib
    from tonic_amd.collector import Block, Collector
    rng = np.random.RandomState(O + W)
    params = _actor(O, A, 3)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in params])).cuda()
    obs = rng.standard_normal((T + 1, W, O)).astype(np.float32)
    eps = rng.standard_normal((T, W, A)).astype(np.float32)
    rewards = rng.standard_normal((T, W)).astype(np.float32)
    resets = (rng.uniform(size=(T, W)) < 0.05).astype(np.float32)
    results, seen = {}, []
    side = torch.cuda.Stream()

    def rollout(transport, company):
        block = Block(W, O, A)
        collector = Collector(block, transport)
        seg = _segment(T, W, O, A)
        sums = torch.zeros(2 * O, device='cuda')
        collector.bind_segment(seg, sums, T)
        torch.cuda.synchronize()
        came, gone = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if company:
            came.record(side)
            _lib.check(lib.tonic_debug_occupy(hog, hog_ms, side.cuda_stream), 'occupy')
            gone.record(side)
            time.sleep(0.001)                      # (the foreign kernel is in before the first command)
        collector.begin_rollout(flat)
        actions, under, clock = [], None, time.perf_counter()
        for t in range(T):
            block.observations[:] = obs[t]
            block.eps[t & 1][:] = eps[t]
            collector.ppo_step(t, t & 1, t > 0)
            collector.wait_actions()
            if company and t in (0, 9):
                now = time.perf_counter()
                print(f'step {t} done {(now - clock) * 1e6:.0f} us after the rollout began, foreign '
                      f'kernel gone: {gone.query()}')
            actions.append(block.actions.copy())
            block.next_observations[:] = obs[t + 1]
            block.rewards[:] = rewards[t]
            block.resets[:] = resets[t]
            block.terminations[:] = 0
            if company and t == min(T, 10) - 1:
                under = not gone.query()           # ten steps done and the foreign kernel still there
        collector.end_rollout(T - 1)
        torch.cuda.synchronize()
        if company:
            print(f'the foreign kernel held its compute units for {came.elapsed_time(gone):.2f} ms')
        out = {k: v.cpu().numpy() for k, v in seg.items()}
        out['sums'] = sums.cpu().numpy()
        out['block_actions'] = np.stack(actions)
        collector.close()
        return out, under

    want, _ = rollout(0, False)
    # (3: the resident kernel with its inputs pushed where a step's observations fit the window's rule —
    #  the 256- and 5-worker cases — and pulled, transport 2, at 1 280 workers: both are covered)
    rollout(3, False)       # (the resident kernel once without company: its first launch in a process
    #                          loads code, which the steps beside the foreign kernel must not wait for)
    for attempt in range(4):
        # EVERY run beside the foreign kernel must give the undisturbed run's bits.  Whether the
        # launch gets compute units while the foreign kernel is there is the hardware dispatcher's
        # choice (workgroups are placed in order, round-robin over the XCDs: when the FIRST one finds
        # its XCD full, none gets in before the foreign kernel leaves — seen in about one run in
        # three): required in one of at most four runs, reported for all.
        got, under = rollout(3, True)
        seen.append(under)
        for key, value in want.items():
            assert np.array_equal(got[key], value), (attempt, key)
        if under:
            break
    print('ten steps done while the foreign kernel held the chip:', seen)
    assert any(seen), 'no run made progress beside the foreign kernel'


@pytest.mark.parametrize('O,A,W,T', [(28, 8, 1280, 500), (17, 6, 256, 1500)])
def test_resident_kernel_under_random_foreign_bursts(lib, request, O, A, W, T):
    """A long rollout while a second thread keeps launching foreign kernels of random width (40 – 250
    compute units) and duration (0.1 – 2 ms) with random pauses: workgroups of the resident kernel are
    absent, arrive, find their slots run, park (the host also sleeps now and then) and are launched
    again in every order the bursts produce.  Segment, normaliser sums and the actions the host read
    must be those of the undisturbed launch-per-step transport, bit for bit."""
    if _in_a_process_of_its_own(request):
        return
    import threading
    import time
    from tonic_amd import _lib
    from tonic_amd.collector import Block, Collector
    rng = np.random.RandomState(7 * O + W)
    params = _actor(O, A, 3)
    flat = torch.as_tensor(np.concatenate([p.reshape(-1) for p in params])).cuda()
    obs = rng.standard_normal((T + 1, W, O)).astype(np.float32)
    eps = rng.standard_normal((T, W, A)).astype(np.float32)
    rewards = rng.standard_normal((T, W)).astype(np.float32)
    resets = (rng.uniform(size=(T, W)) < 0.05).astype(np.float32)
    naps = set(rng.choice(T, size=T // 25, replace=False).tolist())      # host pauses > the park time

    def rollout(transport, bursts):
        block = Block(W, O, A)
        collector = Collector(block, transport)
        seg = _segment(T, W, O, A)
        sums = torch.zeros(2 * O, device='cuda')
        collector.bind_segment(seg, sums, T)
        torch.cuda.synchronize()
        stop, launched = threading.Event(), [0]

        def company():
            side = torch.cuda.Stream()
            noise = np.random.RandomState(11)
            while not stop.is_set():
                _lib.check(lib.tonic_debug_occupy(int(noise.randint(40, 251)),
                                                  float(noise.uniform(0.1, 2.0)), side.cuda_stream),
                           'occupy')
                launched[0] += 1
                time.sleep(float(noise.uniform(0.0, 0.002)))
                side.synchronize()
        thread = threading.Thread(target=company, daemon=True)
        if bursts:
            thread.start()
        collector.begin_rollout(flat)
        actions = []
        for t in range(T):
            block.observations[:] = obs[t]
            block.eps[t & 1][:] = eps[t]
            collector.ppo_step(t, t & 1, t > 0)
            collector.wait_actions()
            actions.append(block.actions.copy())
            block.next_observations[:] = obs[t + 1]
            block.rewards[:] = rewards[t]
            block.resets[:] = resets[t]
            block.terminations[:] = 0
            if bursts and t in naps:
                time.sleep(0.0005)
        collector.end_rollout(T - 1)
        stop.set()
        if bursts:
            thread.join(timeout=10.0)
        torch.cuda.synchronize()
        out = {k: v.cpu().numpy() for k, v in seg.items()}
        out['sums'] = sums.cpu().numpy()
        out['block_actions'] = np.stack(actions)
        collector.close()
        return out, launched[0]

    want, _ = rollout(0, False)
    for attempt in range(2):
        got, launched = rollout(3, True)
        print('foreign kernels launched during the rollout:', launched)
        assert launched >= 5
        for key, value in want.items():
            assert np.array_equal(got[key], value), (attempt, key)


@pytest.mark.parametrize('name,golden_name,prefix', [
    ('PPO', 'ppo_halfcheetah_small', 'init/'), ('SAC', 'sac_small', 'pre/'),
    ('TD3', 'td3_small', 'pre/'), ('DDPG', 'ddpg_small', 'pre/')])
def test_checkpoints_have_exactly_the_reference_state_dict(lib, golden, tmp_path, name,
                                                          golden_name, prefix):
    """Agent.save after the parameters moved into the flat (for SAC / TD3 / DDPG: padded) HBM
    blocks: torch.load gives exactly the key set and shapes of the reference's state_dict
    (recorded from the unmodified reference in the golden), so tonic.play / the reference agents
    load it (tonic/torch/agents/agent.py:17-26)."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    g = golden(golden_name)
    reference = {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}
    if name == 'PPO':
        O = reference['actor.torso.model.0.weight'].shape[1]
        A = reference['actor.head.loc_layer.0.weight'].shape[0]
        agent = tonic_amd.torch.agents.PPO()
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
    else:       # the off-policy goldens were recorded with 32-wide torsos: same model here
        from test_gpu_offpolicy import _agent_from_golden
        agent = _agent_from_golden(g, name.lower())
    agent.save(str(tmp_path / 'step_1'))
    saved = torch.load(tmp_path / 'step_1.pt', map_location='cpu')
    assert set(saved) == set(reference)
    for key, value in saved.items():
        assert tuple(value.shape) == reference[key].shape, key
        assert value.is_contiguous() and value.dtype == torch.float32


def test_push_transport_is_in_effect_for_the_metrics_shapes(lib):
    """transport 3 (command word, observation rows and noise rows pushed into a device window by the host)
    is what a collector asked for it runs on an MI355X for cfg 1 / cfg 5's per-step shapes (W = 256), and
    what a block beyond 48 KB of observations per step, a second collector on the same block and
    TONIC_AMD_COLLECTOR_PUSH=0 fall back from — each to the resident pull transport."""
    import os
    from tonic_amd.collector import Block, Collector
    block = Block(256, 28, 8)
    first = Collector(block, 3)
    assert first.transport == 3, 'this GPU has no CPU-visible device memory?'
    second = Collector(block, 3)                  # the block's window belongs to the first
    assert second.transport == 2
    second.close()
    first.close()
    again = Collector(block, 3)                   # ... and is free again
    assert again.transport == 3
    again.close()
    many = Collector(Block(1280, 28, 8), 3)       # 143 KB of observations per step: the pull is faster
    assert many.transport == 2
    many.close()
    os.environ['TONIC_AMD_COLLECTOR_PUSH'] = '0'
    try:
        off = Collector(Block(256, 28, 8), 3)
        assert off.transport == 2
        off.close()
    finally:
        del os.environ['TONIC_AMD_COLLECTOR_PUSH']
