"""TEST INFRASTRUCTURE ONLY — golden-vector generator.

Runs the *unmodified* reference (``/root/reference``, imported through
``oracle/reference_loader.py``) on seeded synthetic inputs and snapshots inputs and
outputs as ``tests/golden/*.npz``.  ``/root/reference`` does not exist on the GPU box,
so the fixtures (small) are committed together with this script:

    python oracle/make_golden.py            # regenerates every fixture, 1 torch thread

Each fixture records the reference function it came from in its ``source`` field.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_loader as rl  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


class RecordingLogger:
    """Stands in for ``tonic.logger.current_logger`` to capture per-iteration infos."""

    def __init__(self):
        self.records = {}

    def store(self, key, value, stats=False):
        self.records.setdefault(key, []).append(np.array(value))


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def state_arrays(prefix, state_dict):
    return {prefix + k: v.detach().numpy().copy() for k, v in state_dict.items()}


# ----------------------------------------------------------------------------- cases

def golden_lambda_returns(tonic):
    """tonic/replays/utils.py:4-19 and segments.py:41-46 on edge-case shapes."""
    rng = np.random.RandomState(0)
    cases = {}
    shapes = [(1, 1), (1, 7), (5, 1), (64, 8), (33, 5), (128, 3), (256, 2), (512, 4)]
    for i, (steps, workers) in enumerate(shapes):
        nv = rng.normal(size=(steps, workers)).astype(np.float32)
        val = rng.normal(size=(steps, workers)).astype(np.float32)
        rew = rng.normal(size=(steps, workers)).astype(np.float32)
        resets = (rng.uniform(size=(steps, workers)) < 0.15)
        terms = resets & (rng.uniform(size=(steps, workers)) < 0.5)
        if i == 4:      # every transition is a time-out
            resets[:] = True
            terms[:] = False
        if i == 5:      # every transition terminates
            resets[:] = True
            terms[:] = True
        if i == 6:      # nothing ever resets (longest dependency chains)
            resets[:] = False
            terms[:] = False
        resets = resets.astype(np.float32)
        terms = terms.astype(np.float32)
        for j, (gamma, lam) in enumerate([(0.99, 0.97), (0.9, 0.5), (1.0, 1.0), (0.99, 0.0)]):
            ret = tonic.replays.lambda_returns(
                values=val, next_values=nv, rewards=rew, resets=resets,
                terminations=terms, discount_factor=gamma, trace_decay=lam)
            seg = tonic.replays.Segment(size=steps, discount_factor=gamma, trace_decay=lam)
            seg.initialize(0)
            seg.buffers = dict(rewards=rew, resets=resets, terminations=terms)
            seg.compute_returns(val.reshape(-1), nv.reshape(-1))
            adv = seg.get_full('advantages')['advantages']
            assert np.array_equal(seg.buffers['returns'], ret)
            key = f'c{i}_{j}_'
            cases.update({key + 'next_values': nv, key + 'values': val, key + 'rewards': rew,
                          key + 'resets': resets, key + 'terminations': terms,
                          key + 'gamma': np.float64(gamma), key + 'lambda': np.float64(lam),
                          key + 'returns': ret, key + 'advantages': adv.reshape(steps, workers)})
    # constant advantages -> std == 0 -> normalisation skipped (segments.py:44)
    seg = tonic.replays.Segment(size=4)
    seg.initialize(0)
    ones = np.ones((4, 3), np.float32)
    seg.buffers = dict(returns=ones * 2, values=ones)
    cases['const_advantages'] = seg.get_full('advantages')['advantages']
    save('lambda_returns', source='tonic/replays/utils.py:4-19; segments.py:38-48,67-78',
         n_cases=len(shapes), **cases)


def golden_meanstd(tonic):
    """tonic/torch/normalizers/mean_stds.py:44-74."""
    rng = np.random.RandomState(1)
    norm = tonic.torch.normalizers.MeanStd()
    norm.initialize((5,))
    out = {}
    for u in range(3):
        for s in range(4):
            batch = (rng.normal(size=(7, 5)) * (u + 1) + s).astype(np.float32)
            norm.record(batch)
            out[f'u{u}_s{s}_batch'] = batch
        out[f'u{u}_sum'] = np.array(norm.new_sum)
        out[f'u{u}_sum_sq'] = np.array(norm.new_sum_sq)
        norm.update()
        out[f'u{u}_mean'] = norm._mean.detach().numpy().copy()
        out[f'u{u}_std'] = norm._std.detach().numpy().copy()
        x = rng.normal(size=(6, 5)).astype(np.float32)
        out[f'u{u}_x'] = x
        out[f'u{u}_normalized'] = norm(torch.as_tensor(x)).numpy()
    save('meanstd', source='tonic/torch/normalizers/mean_stds.py:34-74', **out)


def golden_buffer(tonic):
    """tonic/replays/buffers.py:28-91 (default return_steps=1)."""
    out = {}
    for case, (workers, size, batch) in enumerate([(1, 50, 8), (4, 64, 16), (3, 20, 5)]):
        rng = np.random.RandomState(10 + case)
        buf = tonic.replays.Buffer(size=size, batch_iterations=3, batch_size=batch,
                                   steps_before_batches=0, steps_between_batches=1)
        buf.initialize(seed=case)
        n_store = size // workers + 5        # wraps the circular index
        pre = f'b{case}_'
        stored = {k: [] for k in ('observations', 'actions', 'next_observations',
                                  'rewards', 'resets', 'terminations')}
        for _ in range(n_store):
            kw = dict(observations=rng.normal(size=(workers, 3)).astype(np.float32),
                      actions=rng.uniform(-1, 1, size=(workers, 2)).astype(np.float32),
                      next_observations=rng.normal(size=(workers, 3)).astype(np.float32),
                      rewards=rng.normal(size=workers).astype(np.float32),
                      resets=rng.uniform(size=workers) < 0.2,
                      terminations=rng.uniform(size=workers) < 0.1)
            for k in stored:
                stored[k].append(kw[k])
            buf.store(**kw)
        for k in stored:
            out[pre + 'in_' + k] = np.array(stored[k])
        out[pre + 'index'] = np.int64(buf.index)
        out[pre + 'size'] = np.int64(buf.size)
        out[pre + 'max_size'] = np.int64(buf.max_size)
        for k, v in buf.buffers.items():
            out[pre + 'buf_' + k] = v.copy()
        state = buf.np_random.get_state()
        keys = ('observations', 'actions', 'next_observations', 'rewards', 'discounts')
        for it, batch_dict in enumerate(buf.get(*keys, steps=123)):
            for k in keys:
                out[pre + f'get{it}_' + k] = batch_dict[k]
        replay_rng = np.random.RandomState()
        replay_rng.set_state(state)
        for it in range(3):
            out[pre + f'get{it}_indices'] = replay_rng.randint(buf.size * workers, size=batch)
        out[pre + 'cfg'] = np.array([workers, size, batch], np.int64)
    save('buffer', source='tonic/replays/buffers.py:28-91', **out)


def golden_buffer_nstep(tonic):
    """tonic/replays/buffers.py:33-79 with return_steps > 1 (accumulate_n_steps)."""
    out = {}
    for case, (workers, size, steps) in enumerate([(4, 48, 3), (1, 9, 5), (5, 40, 2)]):
        rng = np.random.RandomState(40 + case)
        buf = tonic.replays.Buffer(size=size, return_steps=steps)
        buf.initialize(seed=case)
        n_store = size // workers + 7        # wraps the circular index
        pre = f'n{case}_'
        names = ('observations', 'actions', 'next_observations', 'rewards', 'resets',
                 'terminations')
        stored = {k: [] for k in names}
        for t in range(n_store):
            kw = dict(observations=rng.normal(size=(workers, 3)).astype(np.float32),
                      actions=rng.uniform(-1, 1, size=(workers, 2)).astype(np.float32),
                      next_observations=rng.normal(size=(workers, 3)).astype(np.float32),
                      rewards=rng.normal(size=workers).astype(np.float32),
                      resets=rng.uniform(size=workers) < 0.25,
                      terminations=rng.uniform(size=workers) < 0.15)
            for k in names:
                stored[k].append(kw[k])
            buf.store(**kw)
            if t == 3:                       # early snapshot: fewer stored rows than return_steps
                for k, v in buf.buffers.items():
                    out[pre + 'early_' + k] = v.copy()
        for k in names:
            out[pre + 'in_' + k] = np.array(stored[k])
        for k, v in buf.buffers.items():
            out[pre + 'buf_' + k] = v.copy()
        out[pre + 'cfg'] = np.array([workers, size, steps, buf.index, buf.size], np.int64)
    save('buffer_nstep', source='tonic/replays/buffers.py:33-79', **out)


def golden_segment_minibatches(tonic):
    """tonic/replays/segments.py:50-65 with batch_size set (index stream)."""
    seg = tonic.replays.Segment(size=6, batch_iterations=3, batch_size=7)
    seg.initialize(seed=3)
    seg.num_workers = 4
    seg.buffers = dict(idx=np.arange(24, dtype=np.float32).reshape(6, 4))
    batches = [b['idx'].astype(np.int64) for b in seg.get('idx')]
    save('segment_minibatch', source='tonic/replays/segments.py:50-65',
         seed=np.int64(3), size=np.int64(24), batch_size=np.int64(7),
         iterations=np.int64(3), lengths=np.array([len(b) for b in batches]),
         indices=np.concatenate(batches))


def golden_sequential(tonic):
    """tonic/environments/distributed.py:12-58 (time-out vs termination reset logic)
    and the Sequential(P*S) == Parallel(P, S) seed layout (:106-113)."""
    def builder():
        return rl.SyntheticEnvironment(3, 2, max_episode_steps=5)
    env = tonic.environments.distribute(builder, 1, 4)
    env.initialize(seed=7)
    obs = [env.start()]
    rng = np.random.RandomState(0)
    acts, nobs, rews, rsts, terms = [], [], [], [], []
    for _ in range(12):
        a = rng.uniform(-1, 1, size=(4, 2)).astype(np.float32)
        o, infos = env.step(a)
        acts.append(a)
        obs.append(o)
        nobs.append(infos['observations'])
        rews.append(infos['rewards'])
        rsts.append(infos['resets'])
        terms.append(infos['terminations'])
    save('sequential', source='tonic/environments/distributed.py:12-58',
         observations=np.array(obs), actions=np.array(acts),
         next_observations=np.array(nobs), rewards=np.array(rews),
         resets=np.array(rsts), terminations=np.array(terms),
         seed=np.int64(7), max_episode_steps=np.int64(5))


def run_ppo(tonic, name, obs_dim, act_dim, workers, steps, seed, iterations=80,
            reward_scale=1.0, updates=1, batch_size=None, actor_clip=0, critic_clip=0,
            normalizer_clip=None, algorithm='PPO', entropy_coeff=0, torso=None):
    """tonic/torch/agents/{a2c.py:41-73, ppo.py:20-67}: acts with the reference agent on
    a synthetic env for `steps` time steps so the real store/record/update path runs."""
    def builder():
        return rl.SyntheticEnvironment(obs_dim, act_dim, max_episode_steps=7)
    env = tonic.environments.distribute(builder, 1, workers)
    env.initialize(seed=seed)
    clipped = actor_clip > 0 or critic_clip > 0 or normalizer_clip is not None
    kwargs = {}
    if clipped:     # gradient-norm clipping (actors.py:96-98, critics.py:24-25), MeanStd(clip)
        models, norms = tonic.torch.models, tonic.torch.normalizers
        kwargs = dict(
            model=models.ActorCritic(
                actor=models.Actor(encoder=models.ObservationEncoder(),
                                   torso=models.MLP((64, 64), torch.nn.Tanh),
                                   head=models.DetachedScaleGaussianPolicyHead()),
                critic=models.Critic(encoder=models.ObservationEncoder(),
                                     torso=models.MLP((64, 64), torch.nn.Tanh),
                                     head=models.ValueHead()),
                observation_normalizer=norms.MeanStd(clip=normalizer_clip)),
            actor_updater=tonic.torch.updaters.ClippedRatio(gradient_clip=actor_clip),
            critic_updater=tonic.torch.updaters.VRegression(gradient_clip=critic_clip))
    if torso is not None:           # any MLP(sizes, activation) (models/utils.py:4-23)
        sizes, activation = torso
        models, norms = tonic.torch.models, tonic.torch.normalizers
        act = getattr(torch.nn, activation)
        kwargs['model'] = models.ActorCritic(
            actor=models.Actor(encoder=models.ObservationEncoder(),
                               torso=models.MLP(tuple(sizes), act),
                               head=models.DetachedScaleGaussianPolicyHead()),
            critic=models.Critic(encoder=models.ObservationEncoder(),
                                 torso=models.MLP(tuple(sizes), act), head=models.ValueHead()),
            observation_normalizer=norms.MeanStd())
    if algorithm == 'A2C':          # a2c.py:20-127 with StochasticPolicyGradient (actors.py:9-51)
        kwargs['actor_updater'] = tonic.torch.updaters.StochasticPolicyGradient(
            entropy_coeff=entropy_coeff)
    agent = getattr(tonic.torch.agents, algorithm)(
        replay=tonic.replays.Segment(size=steps, batch_iterations=iterations,
                                     batch_size=batch_size), **kwargs)
    agent.initialize(env.observation_space, env.action_space, seed=seed)
    out = state_arrays('init/', agent.model.state_dict())
    out['entropy_coeff'] = np.float64(entropy_coeff)
    out['clips'] = np.array([actor_clip, critic_clip, normalizer_clip or 0], np.float64)
    out['batch_size'] = np.int64(batch_size or 0)
    out['torso_sizes'] = np.array(torso[0] if torso else (64, 64), np.int64)
    out['torso_activation'] = np.array(torso[1] if torso else 'Tanh')
    recorder = RecordingLogger()
    tonic.logger.current_logger = recorder
    observations = env.start()
    eps_all, act_all, lp_all, obs_all = [], [], [], []
    rng = np.random.RandomState(seed + 1)
    for update in range(updates):
        for t in range(steps):
            gen_state = torch.get_rng_state()
            actions = agent.step(observations, t * workers)
            after = torch.get_rng_state()
            torch.set_rng_state(gen_state)
            eps = torch.randn(workers, act_dim).numpy()
            torch.set_rng_state(after)
            eps_all.append(eps)
            act_all.append(actions.copy())
            lp_all.append(agent.last_log_probs.copy())
            obs_all.append(observations.copy())
            observations, infos = env.step(actions)
            # make rewards less trivial and add a few true terminations
            infos['rewards'] = (infos['rewards'] * reward_scale +
                                rng.normal(size=workers)).astype(np.float32)
            term = rng.uniform(size=workers) < 0.05
            infos['terminations'] = term
            infos['resets'] = infos['resets'] | term
            if t == steps - 1:
                seg = {k: v.copy() for k, v in agent.replay.buffers.items()}
                norm = agent.model.observation_normalizer
                pre_state = state_arrays(f'pre{update}/', agent.model.state_dict())
            agent.update(**infos, steps=t * workers)
        # the last store happened inside update(); rebuild the full segment view
        seg = {k: v.copy() for k, v in agent.replay.buffers.items()}
        pre = f'u{update}/'
        for k in ('observations', 'actions', 'next_observations', 'rewards', 'resets',
                  'terminations', 'log_probs', 'values', 'next_values', 'returns',
                  'advantages'):
            out[pre + 'segment/' + k] = seg[k]
        out.update(pre_state)
        out.update(state_arrays(f'post{update}/', agent.model.state_dict()))
        for k, v in recorder.records.items():
            if k == 'critic/v':
                out[pre + 'info/critic/v_mean'] = np.array([x.mean() for x in v])
                out[pre + 'info/critic/v_first'] = v[0]
            else:
                out[pre + 'info/' + k] = np.array(v)
        recorder.records.clear()
        out[pre + 'norm/count'] = np.int64(norm.count)
        if update == 0 and batch_size is None and not clipped and algorithm == 'PPO' and not torso:
            out.update(first_update_probes(tonic, builder, seed, seg, iterations))
    out['act/observations'] = np.array(obs_all)
    out['act/eps'] = np.array(eps_all)
    out['act/actions'] = np.array(act_all)
    out['act/log_probs'] = np.array(lp_all)
    out['cfg'] = np.array([obs_dim, act_dim, workers, steps, seed, iterations, updates],
                          np.int64)
    save(name, source='tonic/torch/agents/a2c.py:41-99; ppo.py:20-67; '
                      'updaters/actors.py:70-112; updaters/critics.py:18-28', **out)


def first_update_probes(tonic, builder, seed, seg, iterations):
    """Two extra reference runs on the first update's batch (fresh agents, same seed ->
    same initial parameters; updaters called as ppo.py:33-46 does):
      * ``iter1/``  parameters after exactly ONE actor+critic iteration (well conditioned:
        the 1e-5 parameter-delta tolerance applies strictly here);
      * ``noise/``  |delta| between the reference on the batch and the reference on a
        sample-PERMUTED batch after all iterations.  Full-batch means are permutation
        invariant, so this is the reference's own float32 summation-order noise after
        `iterations` Adam steps — the floor any independent implementation can reach."""
    flat = {k: tonic.replays.flatten_batch(v) for k, v in seg.items()}
    n = flat['rewards'].shape[0]

    def run(order, iters):
        env = tonic.environments.distribute(builder, 1, 1)
        agent = tonic.torch.agents.PPO()
        agent.initialize(env.observation_space, env.action_space, seed=seed)
        batch = {k: torch.as_tensor(flat[k][order]) for k in
                 ('observations', 'actions', 'advantages', 'log_probs', 'returns')}
        train_actor = True
        for _ in range(iters):
            if train_actor:
                infos = agent.actor_updater(batch['observations'], batch['actions'],
                                            batch['advantages'], batch['log_probs'])
                train_actor = not infos['stop'].numpy()
            agent.critic_updater(batch['observations'], batch['returns'])
        return agent.model.state_dict()

    identity = np.arange(n)
    one = run(identity, 1)
    full = run(identity, iterations)
    shuffled = run(np.random.RandomState(99).permutation(n), iterations)
    out = state_arrays('iter1/', one)
    for k in full:
        out['noise/' + k] = np.abs(full[k].numpy() - shuffled[k].numpy())
        out['probe_full/' + k] = full[k].numpy().copy()
    return out


def run_offpolicy(tonic, name, kind, obs_dim=11, act_dim=3, workers=4, hidden=32, batch=24,
                  iterations=6, seed=0, loop_steps=16, atoms=(-6.0, 6.0, 21), return_steps=1,
                  samples=4, torso=None):
    """tonic/torch/agents/{ddpg.py:45-112, td3.py:38-55, sac.py:40-51} driven through the
    reference agent on a synthetic env (small custom torso so the fixture stays small).  The
    first learner update is captured completely: buffer contents, the index stream of
    Buffer.get, the standard-normal draws of the updaters (regenerated from the saved torch
    generator state and checked), per-iteration infos, parameters before / after."""
    models, updaters = tonic.torch.models, tonic.torch.updaters
    relu = torch.nn.ReLU
    sizes = (hidden, hidden)
    if torso is not None:           # any MLP(sizes, activation) (models/utils.py:4-23)
        sizes, relu = tuple(torso[0]), getattr(torch.nn, torso[1])

    def builder():
        return rl.SyntheticEnvironment(obs_dim, act_dim, max_episode_steps=5)
    env = tonic.environments.distribute(builder, 1, workers)
    env.initialize(seed=seed)
    critic_head = (models.DistributionalValueHead(*atoms) if kind == 'd4pg'     # d4pg.py:15-17
                   else models.ValueHead())
    critic = models.Critic(encoder=models.ObservationActionEncoder(),
                           torso=models.MLP(sizes, relu), head=critic_head)
    if kind == 'sac':
        head = models.GaussianPolicyHead(loc_activation=torch.nn.Identity,
                                         distribution=models.SquashedMultivariateNormalDiag)
    elif kind == 'mpo':
        head = models.GaussianPolicyHead()                              # mpo.py:12
    else:
        head = models.DeterministicPolicyHead()
    container = (models.ActorCriticWithTargets if kind in ('ddpg', 'd4pg', 'mpo')
                 else models.ActorTwinCriticWithTargets)
    model = container(
        actor=models.Actor(encoder=models.ObservationEncoder(),
                           torso=models.MLP(sizes, relu), head=head),
        critic=critic, observation_normalizer=tonic.torch.normalizers.MeanStd())
    replay = tonic.replays.Buffer(size=400, batch_iterations=iterations, batch_size=batch,
                                  steps_before_batches=workers * 10, steps_between_batches=workers * 10,
                                  return_steps=return_steps)
    if kind == 'sac':
        agent = tonic.torch.agents.SAC(
            model=model, replay=replay,
            exploration=tonic.explorations.NoActionNoise(start_steps=workers * 5))
    elif kind == 'mpo':
        agent = tonic.torch.agents.MPO(
            model=model, replay=replay,
            actor_updater=updaters.MaximumAPosterioriPolicyOptimization(num_samples=samples),
            critic_updater=updaters.ExpectedSARSA(num_samples=samples))
    else:
        cls = {'ddpg': tonic.torch.agents.DDPG, 'd4pg': tonic.torch.agents.D4PG,
               'td3': tonic.torch.agents.TD3}[kind]
        agent = cls(
            model=model, replay=replay,
            exploration=tonic.explorations.NormalActionNoise(start_steps=workers * 5))
    agent.initialize(env.observation_space, env.action_space, seed=seed)
    out = state_arrays('init/', agent.model.state_dict())
    recorder = RecordingLogger()
    tonic.logger.current_logger = recorder
    rng = np.random.RandomState(seed + 1)
    observations = env.start()
    obs_all, act_all, policy_eps = [], [], []
    captured = {}
    original_update = agent._update

    def capturing_update(steps):
        if not captured:
            captured['torch_state'] = torch.get_rng_state()
            captured['np_state'] = agent.replay.np_random.get_state()
            captured['size'] = agent.replay.size
            captured['buffers'] = {k: v.copy() for k, v in agent.replay.buffers.items()}
            captured['pre'] = state_arrays('pre/', agent.model.state_dict())
            original_update(steps)
            captured['post'] = state_arrays('post/', agent.model.state_dict())
            captured['infos'] = {k: np.array(v) for k, v in recorder.records.items()}
        else:
            original_update(steps)
    agent._update = capturing_update
    for t in range(loop_steps):
        gen_state = torch.get_rng_state()
        actions = agent.step(observations, t * workers)
        after = torch.get_rng_state()
        torch.set_rng_state(gen_state)
        policy_eps.append(torch.randn(workers, act_dim).numpy())      # consumed only by SAC policy
        torch.set_rng_state(after)
        obs_all.append(observations.copy())
        act_all.append(np.array(actions, np.float64))
        observations, infos = env.step(actions)
        infos['rewards'] = (infos['rewards'] + rng.normal(size=workers)).astype(np.float32)
        term = rng.uniform(size=workers) < 0.1
        infos['terminations'] = term
        infos['resets'] = infos['resets'] | term
        agent.update(**infos, steps=t * workers)
    assert captured, 'the learner update never ran'
    # regenerate the index stream and the updaters' normal draws, then verify them by replaying
    # the reference updaters on a fresh agent (see the consistency check below)
    index_rng = np.random.RandomState()
    index_rng.set_state(captured['np_state'])
    indices = np.array([index_rng.randint(captured['size'] * workers, size=batch)
                        for _ in range(iterations)])
    saved = torch.get_rng_state()
    torch.set_rng_state(captured['torch_state'])
    draws = 2 if kind in ('sac', 'mpo') else 1
    if kind == 'mpo':       # rsample((S,)) of the critic step, sample((S,)) of the actor step
        eps = np.array([[torch.randn(samples, batch, act_dim).numpy().reshape(-1, act_dim)
                         for _ in range(draws)] for _ in range(iterations)])
    else:
        eps = np.array([[torch.randn(batch, act_dim).numpy() for _ in range(draws)]
                        for _ in range(iterations)])
    torch.set_rng_state(saved)
    out.update(captured['pre'])
    out.update(captured['post'])
    for k, v in captured['buffers'].items():
        out['buffer/' + k] = v
    for k, v in captured['infos'].items():
        if k.startswith('actor/alpha'):
            out['info/' + k] = v
        elif v.ndim == 2:
            out['info/' + k + '_mean'] = v.mean(axis=1)
        else:
            out['info/' + k] = v
    out['indices'] = indices
    out['eps'] = eps
    out['act/observations'] = np.array(obs_all)
    out['act/actions'] = np.array(act_all)
    out['act/policy_eps'] = np.array(policy_eps)
    out['buffer_size'] = np.int64(captured['size'])
    out['torso_sizes'] = np.array(sizes, np.int64)
    out['torso_activation'] = np.array(torso[1] if torso else 'ReLU')
    out['cfg'] = np.array([obs_dim, act_dim, workers, hidden, batch, iterations, seed,
                           loop_steps], np.int64)
    out['atoms'] = np.array(atoms, np.float64)
    out['return_steps'] = np.int64(return_steps)
    out['samples'] = np.int64(samples)
    save(name, source='tonic/torch/agents/ddpg.py:45-112; td3.py:38-55; sac.py:40-51; '
                      'updaters/critics.py:125-235; updaters/actors.py:159-267; '
                      'replays/buffers.py:28-91', **out)


def main():
    torch.set_num_threads(1)
    tonic = rl.load_reference()
    if len(sys.argv) > 1:                    # regenerate only the named goldens
        for name in sys.argv[1:]:
            if name == 'mpo_small':
                run_offpolicy(tonic, 'mpo_small', 'mpo', obs_dim=9, act_dim=3, workers=3,
                              batch=20, seed=11, return_steps=2, samples=4)
            elif name == 'd4pg_small':
                run_offpolicy(tonic, 'd4pg_small', 'd4pg', obs_dim=8, act_dim=3, workers=3,
                              batch=20, seed=7, return_steps=3)
            elif name == 'ddpg_small':
                run_offpolicy(tonic, 'ddpg_small', 'ddpg', obs_dim=7, act_dim=2, workers=2,
                              batch=16, seed=5)
            elif name == 'ppo_clipped_small':
                run_ppo(tonic, 'ppo_clipped_small', 17, 6, workers=8, steps=24, seed=8,
                        iterations=12, updates=2, actor_clip=0.05, critic_clip=0.3,
                        normalizer_clip=1.5)
            elif name == 'a2c_small':
                run_ppo(tonic, 'a2c_small', 17, 6, workers=8, steps=24, seed=9, iterations=6,
                        updates=2, algorithm='A2C', entropy_coeff=0.01)
            elif name == 'trpo_small':
                run_ppo(tonic, 'trpo_small', 17, 6, workers=8, steps=24, seed=13, iterations=6,
                        updates=2, algorithm='TRPO', reward_scale=2.0)
            elif name == 'ppo_ant_wide':
                run_ppo(tonic, 'ppo_ant_wide', 111, 8, workers=6, steps=16, seed=11, iterations=8,
                        updates=1)
            elif name == 'ppo_humanoid_wide':
                run_ppo(tonic, 'ppo_humanoid_wide', 376, 17, workers=4, steps=12, seed=12,
                        iterations=6, updates=1)
            elif name == 'ppo_relu256_small':
                run_ppo(tonic, 'ppo_relu256_small', 17, 6, workers=8, steps=24, seed=21,
                        iterations=10, updates=1, torso=((256, 256), 'ReLU'))
            elif name == 'ppo_tanh3_small':
                run_ppo(tonic, 'ppo_tanh3_small', 11, 3, workers=6, steps=20, seed=22,
                        iterations=8, updates=1, torso=((96, 48, 32), 'Tanh'))
            elif name == 'sac_uneven_small':
                run_offpolicy(tonic, 'sac_uneven_small', 'sac', obs_dim=11, act_dim=3, workers=4,
                              batch=24, seed=23, torso=((100, 60), 'ReLU'))
            elif name == 'td3_elu_small':
                run_offpolicy(tonic, 'td3_elu_small', 'td3', obs_dim=9, act_dim=4, workers=3,
                              batch=20, seed=24, torso=((48, 40), 'ELU'))
            elif name == 'ppo_halfcheetah_w256':
                run_ppo(tonic, 'ppo_halfcheetah_w256', 17, 6, workers=256, steps=3, seed=6,
                        updates=1)
            else:
                globals()['golden_' + name](tonic)
        return
    golden_buffer_nstep(tonic)
    golden_lambda_returns(tonic)
    golden_meanstd(tonic)
    golden_buffer(tonic)
    golden_segment_minibatches(tonic)
    golden_sequential(tonic)
    # cfg-2 shapes (HalfCheetah O=17, A=6) at N = 32*8 = 256, two consecutive updates.
    run_ppo(tonic, 'ppo_halfcheetah_small', 17, 6, workers=8, steps=32, seed=0, updates=2)
    # cfg-1 shapes (Pendulum O=3, A=1), W=1.
    run_ppo(tonic, 'ppo_pendulum_small', 3, 1, workers=1, steps=64, seed=1, updates=1)
    # cfg-5 shapes (AntBullet O=28, A=8), larger rewards so the KL stop triggers.
    run_ppo(tonic, 'ppo_antbullet_small', 28, 8, workers=16, steps=24, seed=2,
            reward_scale=5.0, updates=1)
    # minibatch mode (segments.py:58-65): N = 20*12 = 240 samples, ragged last minibatch of 48
    run_ppo(tonic, 'ppo_minibatch_small', 17, 6, workers=12, steps=20, seed=4, iterations=5,
            batch_size=64)
    # the metric's worker count (parallel=256, BASELINE cfg 2) on a short segment: N = 3*256
    run_ppo(tonic, 'ppo_halfcheetah_w256', 17, 6, workers=256, steps=3, seed=6, updates=1)
    # gradient_clip on both updaters and a clipping observation normaliser; two updates so that
    # the second one runs with non-trivial normaliser statistics (values beyond +-1.5 exist)
    run_ppo(tonic, 'ppo_clipped_small', 17, 6, workers=8, steps=24, seed=8, iterations=12,
            updates=2, actor_clip=0.05, critic_clip=0.3, normalizer_clip=1.5)
    # shapes beyond the fused kernels: Ant-v3 (O = 111, A = 8) and Humanoid-v3 (O = 376, A = 17)
    run_ppo(tonic, 'ppo_ant_wide', 111, 8, workers=6, steps=16, seed=11, iterations=8, updates=1)
    run_ppo(tonic, 'ppo_humanoid_wide', 376, 17, workers=4, steps=12, seed=12, iterations=6,
            updates=1)
    # A2C (StochasticPolicyGradient with an entropy bonus): one actor step + 6 critic steps
    run_ppo(tonic, 'a2c_small', 17, 6, workers=8, steps=24, seed=9, iterations=6, updates=2,
            algorithm='A2C', entropy_coeff=0.01)
    # TRPO (trpo.py:7-97, actors.py:115-156, optimizers.py:25-115): conjugate gradient + backtracking
    run_ppo(tonic, 'trpo_small', 17, 6, workers=8, steps=24, seed=13, iterations=6, updates=2,
            algorithm='TRPO', reward_scale=2.0)
    # torsos outside the hand-written kernels' shapes (stock torch operators on the device):
    # PPO with MLP((256, 256), ReLU) and with three tanh layers, SAC with unequal ReLU layers
    # (100, 60: the (400, 300) class at fixture size), TD3 with unequal ELU layers
    run_ppo(tonic, 'ppo_relu256_small', 17, 6, workers=8, steps=24, seed=21, iterations=10,
            updates=1, torso=((256, 256), 'ReLU'))
    run_ppo(tonic, 'ppo_tanh3_small', 11, 3, workers=6, steps=20, seed=22, iterations=8, updates=1,
            torso=((96, 48, 32), 'Tanh'))
    run_offpolicy(tonic, 'sac_uneven_small', 'sac', obs_dim=11, act_dim=3, workers=4, batch=24,
                  seed=23, torso=((100, 60), 'ReLU'))
    run_offpolicy(tonic, 'td3_elu_small', 'td3', obs_dim=9, act_dim=4, workers=3, batch=20, seed=24,
                  torso=((48, 40), 'ELU'))
    run_offpolicy(tonic, 'sac_small', 'sac')
    run_offpolicy(tonic, 'td3_small', 'td3', obs_dim=9, act_dim=4, workers=3, batch=20, seed=3)
    run_offpolicy(tonic, 'ddpg_small', 'ddpg', obs_dim=7, act_dim=2, workers=2, batch=16, seed=5)
    # D4PG (d4pg.py:21-37): 21-atom distributional critic, 3-step returns
    run_offpolicy(tonic, 'd4pg_small', 'd4pg', obs_dim=8, act_dim=3, workers=3, batch=20, seed=7,
                  return_steps=3)
    # MPO (mpo.py:21-109): ExpectedSARSA + the MPO actor / dual step, 4 sampled actions per state
    run_offpolicy(tonic, 'mpo_small', 'mpo', obs_dim=9, act_dim=3, workers=3, batch=20, seed=11,
                  return_steps=2, samples=4)


if __name__ == '__main__':
    main()
