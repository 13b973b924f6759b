"""Pins the CPU oracle (oracle/numpy_port.py) against golden vectors produced by the
unmodified reference (oracle/make_golden.py).  Runs on CPU, no reference checkout needed."""
import numpy as np
import pytest

import numpy_port as port

PPO_CASES = ['ppo_halfcheetah_small', 'ppo_pendulum_small', 'ppo_antbullet_small',
             'ppo_halfcheetah_w256', 'ppo_ant_wide', 'ppo_humanoid_wide']


def test_lambda_returns_bit_exact(golden):
    g = golden('lambda_returns')
    for i in range(int(g['n_cases'])):
        for j in range(4):
            k = f'c{i}_{j}_'
            ret = port.lambda_returns(g[k + 'next_values'], g[k + 'rewards'], g[k + 'resets'],
                                      g[k + 'terminations'], float(g[k + 'gamma']),
                                      float(g[k + 'lambda']))
            assert np.array_equal(ret, g[k + 'returns']), k
            adv = port.normalized_advantages(ret, g[k + 'values'])
            assert np.array_equal(adv, g[k + 'advantages'], equal_nan=True), k
            # affine form (what the chunked HIP scan composes) stays within fp32 rounding
            aff = port.lambda_returns_affine(g[k + 'next_values'], g[k + 'rewards'],
                                             g[k + 'resets'], g[k + 'terminations'],
                                             float(g[k + 'gamma']), float(g[k + 'lambda']))
            scale = max(1.0, np.abs(aff).max())
            assert np.abs(aff - ret).max() <= 2e-5 * scale, k
    const = port.normalized_advantages(np.full((4, 3), 2, np.float32), np.ones((4, 3), np.float32))
    assert np.array_equal(const.reshape(-1), g['const_advantages'])


def test_meanstd_bit_exact(golden):
    g = golden('meanstd')
    norm = port.MeanStdPort((5,))
    for u in range(3):
        for s in range(4):
            norm.record(g[f'u{u}_s{s}_batch'])
        assert np.array_equal(norm.new_sum, g[f'u{u}_sum'])
        assert np.array_equal(norm.new_sum_sq, g[f'u{u}_sum_sq'])
        mean, std = norm.update()
        assert np.array_equal(mean, g[f'u{u}_mean'])
        assert np.array_equal(std, g[f'u{u}_std'])
        assert np.array_equal(norm.normalize(g[f'u{u}_x']), g[f'u{u}_normalized'])


def test_buffer_index_math_bit_exact(golden):
    g = golden('buffer')
    for case in range(3):
        pre = f'b{case}_'
        workers, size, batch = (int(x) for x in g[pre + 'cfg'])
        rng = np.random.RandomState(case)
        n_store = g[pre + 'in_rewards'].shape[0]
        max_size = size // workers
        assert max_size == int(g[pre + 'max_size'])
        assert n_store % max_size == int(g[pre + 'index'])
        disc = port.buffer_discounts(g[pre + 'in_terminations'][-1], 0.99)
        row = (n_store - 1) % max_size
        assert np.array_equal(disc, g[pre + 'buf_discounts'][row])
        cur = min(n_store, max_size)
        for it in range(3):
            flat, rows, cols = port.buffer_sample_indices(rng, cur, workers, batch)
            assert np.array_equal(flat, g[pre + f'get{it}_indices'])
            for key in ('observations', 'actions', 'next_observations', 'rewards', 'discounts'):
                assert np.array_equal(g[pre + 'buf_' + key][rows, cols],
                                      g[pre + f'get{it}_' + key], equal_nan=True)


def nstep_cases(g):
    for case in range(3):
        pre = f'n{case}_'
        workers, size, steps, index, filled = (int(x) for x in g[pre + 'cfg'])
        rows = [{k: g[pre + 'in_' + k][t] for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations')}
            for t in range(g[pre + 'in_rewards'].shape[0])]
        yield pre, workers, size, steps, index, filled, rows


def test_buffer_n_step_accumulation_bit_exact(golden):
    """Buffer(return_steps > 1): store + accumulate_n_steps (buffers.py:33-79)."""
    g = golden('buffer_nstep')
    for pre, workers, size, steps, index, filled, rows in nstep_cases(g):
        buf = port.BufferPort(size, workers, return_steps=steps)
        for t, row in enumerate(rows):
            buf.store(**row)
            if t == 3:
                for k in buf.KEYS:
                    assert np.array_equal(buf.buffers[k], g[pre + 'early_' + k], equal_nan=True), k
        assert (buf.index, buf.size) == (index, filled)
        for k in buf.KEYS:
            assert np.array_equal(buf.buffers[k], g[pre + 'buf_' + k], equal_nan=True), k


def test_segment_minibatch_indices_bit_exact(golden):
    g = golden('segment_minibatch')
    rng = np.random.RandomState(int(g['seed']))
    got = np.concatenate(list(port.segment_minibatch_indices(
        rng, int(g['size']), int(g['batch_size']), int(g['iterations']))))
    assert np.array_equal(got, g['indices'])


def _params(g, prefix):
    actor = [g[prefix + 'actor.torso.model.0.weight'], g[prefix + 'actor.torso.model.0.bias'],
             g[prefix + 'actor.torso.model.2.weight'], g[prefix + 'actor.torso.model.2.bias'],
             g[prefix + 'actor.head.log_scale'], g[prefix + 'actor.head.loc_layer.0.weight'],
             g[prefix + 'actor.head.loc_layer.0.bias']]
    critic = [g[prefix + 'critic.torso.model.0.weight'], g[prefix + 'critic.torso.model.0.bias'],
              g[prefix + 'critic.torso.model.2.weight'], g[prefix + 'critic.torso.model.2.bias'],
              g[prefix + 'critic.head.v_layer.weight'], g[prefix + 'critic.head.v_layer.bias']]
    norm = (g[prefix + 'observation_normalizer._mean'], g[prefix + 'observation_normalizer._std'])
    return actor, critic, norm


@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_acting_matches_reference(golden, name):
    g = golden(name)
    actor, _, _ = _params(g, 'init/')
    steps = int(g['cfg'][3])
    for t in range(steps):      # first update window uses the initial parameters
        act, lp = port.ppo_act(actor, g['act/observations'][t], g['act/eps'][t])
        np.testing.assert_allclose(act, g['act/actions'][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose(lp, g['act/log_probs'][t], rtol=0, atol=2e-5)


@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_update_matches_reference(golden, name):
    g = golden(name)
    updates = int(g['cfg'][6])
    actor_adam = critic_adam = None
    for u in range(updates):
        actor, critic, norm = _params(g, f'pre{u}/')
        seg = {k: g[f'u{u}/segment/{k}'] for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets',
            'terminations', 'log_probs')}
        if actor_adam is None:
            actor_adam = port.AdamPort(actor, 3e-4)
            critic_adam = port.AdamPort(critic, 1e-3)
        new_actor, new_critic, infos, extra = port.ppo_update(
            actor, critic, norm, seg, batch_iterations=int(g['cfg'][5]), actor_adam=actor_adam,
            critic_adam=critic_adam)
        np.testing.assert_allclose(extra['returns'], g[f'u{u}/segment/returns'], atol=1e-5, rtol=1e-6)
        np.testing.assert_allclose(extra['advantages'].reshape(seg['rewards'].shape),
                                   g[f'u{u}/segment/advantages'], atol=1e-5, rtol=1e-5)
        n_actor = int(g[f'u{u}/info/actor/iterations'][0])
        assert sum('actor' in i for i in infos) == n_actor
        for key in ('loss', 'kl', 'entropy', 'clip_fraction', 'std'):
            got = np.array([i['actor'][key] for i in infos if 'actor' in i])
            np.testing.assert_allclose(got, g[f'u{u}/info/actor/{key}'], atol=1e-5, rtol=1e-5)
        stops = np.array([i['actor']['stop'] for i in infos if 'actor' in i])
        assert np.array_equal(stops, g[f'u{u}/info/actor/stop'])
        closs = np.array([i['critic']['loss'] for i in infos])
        np.testing.assert_allclose(closs, g[f'u{u}/info/critic/loss'], rtol=1e-5, atol=1e-5)
        vmean = np.array([i['critic']['v'].mean() for i in infos])
        np.testing.assert_allclose(vmean, g[f'u{u}/info/critic/v_mean'], rtol=1e-5, atol=1e-5)
        ref_actor, ref_critic, _ = _params(g, f'post{u}/')
        noise = _params(g, 'noise/')
        for got, want, before, floor in zip(new_actor + new_critic, ref_actor + ref_critic,
                                            actor + critic, noise[0] + noise[1]):
            # Parameter *deltas* within 1e-5 (north_star tolerance) after ALL iterations,
            # unless the reference itself moves by more than that when only its float32
            # summation order changes (golden 'noise/': reference vs reference on a
            # sample-permuted batch) — then the bound is a multiple of that floor.
            tol = max(1e-5, 50 * float(floor.max()))
            np.testing.assert_allclose(got - before, want - before, atol=tol, rtol=0)


def test_ppo_update_with_gradient_and_normaliser_clipping(golden):
    """`gradient_clip` on both updaters (actors.py:96-98, critics.py:24-25) and
    `MeanStd(clip=1.5)` (mean_stds.py:37-38): two consecutive reference updates, the second one
    with learnt normaliser statistics so that inputs beyond the clip exist."""
    g = golden('ppo_clipped_small')
    actor_clip, critic_clip, normalizer_clip = (float(x) for x in g['clips'])
    actor_adam = critic_adam = None
    for u in range(int(g['cfg'][6])):
        actor, critic, norm = _params(g, f'pre{u}/')
        seg = {k: g[f'u{u}/segment/{k}'] for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets',
            'terminations', 'log_probs')}
        if actor_adam is None:
            actor_adam, critic_adam = port.AdamPort(actor, 3e-4), port.AdamPort(critic, 1e-3)
        new_actor, new_critic, infos, extra = port.ppo_update(
            actor, critic, norm, seg, batch_iterations=int(g['cfg'][5]), actor_adam=actor_adam,
            critic_adam=critic_adam, actor_clip=actor_clip, critic_clip=critic_clip,
            normalizer_clip=normalizer_clip)
        np.testing.assert_allclose(extra['returns'], g[f'u{u}/segment/returns'], atol=1e-5, rtol=1e-6)
        n_actor = int(g[f'u{u}/info/actor/iterations'][0])
        assert sum('actor' in i for i in infos) == n_actor
        kl = np.array([i['actor']['kl'] for i in infos if 'actor' in i])
        np.testing.assert_allclose(kl, g[f'u{u}/info/actor/kl'], atol=1e-5, rtol=1e-5)
        closs = np.array([i['critic']['loss'] for i in infos])
        np.testing.assert_allclose(closs, g[f'u{u}/info/critic/loss'], rtol=1e-5, atol=1e-5)
        ref_actor, ref_critic, _ = _params(g, f'post{u}/')
        for got, want, before in zip(new_actor + new_critic, ref_actor + ref_critic, actor + critic):
            np.testing.assert_allclose(got - before, want - before, atol=2e-5, rtol=0)
    # the clipping must actually bite in this golden
    x = (g['u1/segment/observations'] - norm[0]) / norm[1]
    assert (np.abs(x) > normalizer_clip).mean() > 0.01


def test_a2c_update_matches_reference(golden):
    """A2C (a2c.py:101-127): ONE StochasticPolicyGradient step with an entropy bonus on the full
    batch, then `batch_iterations` VRegression steps; two consecutive reference updates."""
    g = golden('a2c_small')
    entropy_coeff, iterations = float(g['entropy_coeff']), int(g['cfg'][5])
    actor_adam = critic_adam = None
    for u in range(int(g['cfg'][6])):
        actor, critic, norm = _params(g, f'pre{u}/')
        seg = {k: g[f'u{u}/segment/{k}'] for k in (
            'observations', 'actions', 'log_probs', 'advantages', 'returns')}
        flat = {k: port.flatten_time_major(v) for k, v in seg.items()}
        if actor_adam is None:
            actor_adam, critic_adam = port.AdamPort(actor, 3e-4), port.AdamPort(critic, 1e-3)
        grads, stats = port.clipped_ratio_grads(
            actor, flat['observations'], flat['actions'], flat['advantages'].reshape(-1),
            flat['log_probs'], entropy_coeff=entropy_coeff, plain=True)
        new_actor = actor_adam.step(actor, grads)
        for key in ('loss', 'kl', 'entropy', 'std'):
            np.testing.assert_allclose(stats[key], g[f'u{u}/info/actor/{key}'][0], rtol=1e-5, atol=1e-5)
        new_critic, losses = critic, []
        for _ in range(iterations):
            grads, stats = port.value_regression_grads(new_critic, norm[0], norm[1],
                                                       flat['observations'], flat['returns'].reshape(-1))
            new_critic = critic_adam.step(new_critic, grads)
            losses.append(stats['loss'])
        np.testing.assert_allclose(losses, g[f'u{u}/info/critic/loss'], rtol=1e-5, atol=1e-5)
        ref_actor, ref_critic, _ = _params(g, f'post{u}/')
        for got, want, before in zip(new_actor + new_critic, ref_actor + ref_critic, actor + critic):
            np.testing.assert_allclose(got - before, want - before, atol=1e-5, rtol=0)


@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_single_iteration_deltas_strict(golden, name):
    """One actor + one critic optimizer step: parameter deltas within 1e-5, strictly."""
    g = golden(name)
    actor, critic, norm = _params(g, 'pre0/')
    seg = {k: g[f'u0/segment/{k}'] for k in (
        'observations', 'actions', 'next_observations', 'rewards', 'resets',
        'terminations', 'log_probs')}
    new_actor, new_critic, _, extra = port.ppo_update(actor, critic, norm, seg, batch_iterations=1)
    ref_actor, ref_critic, _ = _params(g, 'iter1/')
    flat = {k: port.flatten_time_major(v) for k, v in seg.items()}
    g_actor, _ = port.clipped_ratio_grads(actor, flat['observations'], flat['actions'],
                                          extra['advantages'], flat['log_probs'])
    g_critic, _ = port.value_regression_grads(critic, norm[0], norm[1], flat['observations'],
                                              extra['returns'].reshape(-1))
    for got, want, before, grad in zip(new_actor + new_critic, ref_actor + ref_critic,
                                       actor + critic, g_actor + g_critic):
        # Adam's first step is lr * g / (|g| + 1e-8): where |g| is at float32 summation-noise
        # level the reference's own step has an arbitrary sign, so only the bound |step| <= lr
        # can be checked there; everywhere else the 1e-5 tolerance applies strictly.
        live = np.abs(grad) > 1e-6 * np.abs(grad).max()
        assert live.mean() > 0.99
        np.testing.assert_allclose((got - before)[live], (want - before)[live], atol=1e-5, rtol=0)
        assert np.abs(got - before).max() <= 1.01e-3


@pytest.mark.parametrize('name', ['ppo_halfcheetah_small', 'ppo_pendulum_small'])
def test_torch_port_matches_reference(golden, name):
    """oracle/torch_port.py (the timed CPU baseline) reproduces the reference's update."""
    import torch
    import torch_port
    torch.set_num_threads(1)
    g = golden(name)
    O, A, W, steps = (int(x) for x in g['cfg'][:4])
    actor, critic, norm = _params(g, 'pre0/')
    agent = torch_port.TorchPPO(O, A, steps=steps)
    agent.load(actor, critic, norm)
    agent.buffers = {k: g[f'u0/segment/{k}'].copy() for k in (
        'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
        'log_probs')}
    agent.normalizer.new_count = 1
    agent.normalizer.new_sum = np.zeros(O, np.float32)
    agent.normalizer.new_sum_sq = np.zeros(O, np.float32)
    infos = agent.update()
    assert np.array_equal(agent.buffers['returns'], g['u0/segment/returns'])
    kl = np.array([float(i['actor']['kl']) for i in infos if 'actor' in i])
    np.testing.assert_allclose(kl, g['u0/info/actor/kl'], rtol=1e-6, atol=1e-7)
    closs = np.array([float(i['critic']['loss']) for i in infos])
    np.testing.assert_allclose(closs, g['u0/info/critic/loss'], rtol=1e-6, atol=1e-7)
    ref_actor, ref_critic, _ = _params(g, 'post0/')
    for got, want in zip(agent.actor_vars + agent.critic_vars, ref_actor + ref_critic):
        np.testing.assert_allclose(got.detach().numpy(), want, rtol=0, atol=1e-6)


def test_trpo_port_matches_reference(golden):
    """oracle/torch_port.TorchTRPO — conjugate gradient on Fisher-vector products, backtracking
    line search, critic regression — against two consecutive updates of the reference's TRPO agent."""
    import torch
    import torch_port
    torch.set_num_threads(1)
    g = golden('trpo_small')
    O, A, W, steps, seed, iterations, updates = (int(x) for x in g['cfg'])
    agent = torch_port.TorchTRPO(O, A, steps=steps, iterations=iterations)
    for u in range(updates):
        actor, critic, norm = _params(g, f'pre{u}/')
        agent.load(actor, critic, norm)
        agent.buffers = {k: g[f'u{u}/segment/{k}'].copy() for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
            'log_probs')}
        agent.normalizer.new_count = 1
        agent.normalizer.new_sum = np.zeros(O, np.float32)
        agent.normalizer.new_sum_sq = np.zeros(O, np.float32)
        info = agent.update()
        assert np.array_equal(agent.buffers['returns'], g[f'u{u}/segment/returns'])
        assert info['actor']['backtrack_steps'] == int(g[f'u{u}/info/actor/backtrack_steps'][0])
        np.testing.assert_allclose(info['actor']['loss'], g[f'u{u}/info/actor/loss'][0], rtol=1e-5,
                                   atol=1e-7)
        np.testing.assert_allclose(info['actor']['kl'], g[f'u{u}/info/actor/kl'][0], rtol=1e-4,
                                   atol=1e-7)
        np.testing.assert_allclose([float(c['loss']) for c in info['critic']],
                                   g[f'u{u}/info/critic/loss'], rtol=1e-6, atol=1e-6)
        ref_actor, ref_critic, _ = _params(g, f'post{u}/')
        for got, want in zip(agent.actor_vars + agent.critic_vars, ref_actor + ref_critic):
            np.testing.assert_allclose(got.detach().numpy(), want, rtol=0, atol=2e-6)


def test_c_restatement_matches_golden(golden):
    """oracle/gae_ref.c (built by __graft_entry__.build()) is bit-exact with the reference."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle',
                        'liboracle_gae.so')
    if not os.path.exists(path):
        pytest.skip('oracle C checker not built (run python __graft_entry__.py build)')
    lib = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_lambda_returns.argtypes = [fp] * 5 + [ctypes.c_int64] * 2 + [ctypes.c_double] * 2
    g = golden('lambda_returns')
    for i in range(int(g['n_cases'])):
        for j in range(4):
            k = f'c{i}_{j}_'
            arrays = [np.ascontiguousarray(g[k + n]) for n in
                      ('next_values', 'rewards', 'resets', 'terminations')]
            out = np.zeros_like(arrays[0])
            T, W = out.shape
            lib.oracle_lambda_returns(*[a.ctypes.data_as(fp) for a in arrays],
                                      out.ctypes.data_as(fp), T, W, float(g[k + 'gamma']),
                                      float(g[k + 'lambda']))
            assert np.array_equal(out, g[k + 'returns']), k


@pytest.mark.parametrize('name,kind', [('sac_small', 'sac'), ('td3_small', 'td3'),
                                       ('ddpg_small', 'ddpg'), ('d4pg_small', 'd4pg'),
                                       ('mpo_small', 'mpo')])
def test_offpolicy_port_matches_reference(golden, name, kind):
    """oracle/torch_port.OffPolicyPort replays the reference's first SAC / TD3 update from the
    golden buffer, index stream and normal draws (this also pins the RNG bookkeeping)."""
    import torch
    import torch_port
    torch.set_num_threads(1)
    g = golden(name)
    workers = int(g['cfg'][2])
    state = {k: g[k] for k in g.files}
    port_ = torch_port.OffPolicyPort(kind, state, 'pre/',
                                     atoms=g['atoms'] if kind == 'd4pg' else None,
                                     samples=int(g['samples']) if kind == 'mpo' else 20)
    buffers = {k[len('buffer/'):]: g[k] for k in g.files if k.startswith('buffer/')}
    # Buffer.store semantics (buffers.py:33-56): discounts, NaN padding beyond `size`
    size = int(g['buffer_size'])
    if kind not in ('d4pg', 'mpo'):   # (those: n-step returns, discounts accumulated, buffers.py:58-79)
        assert np.array_equal(buffers['discounts'][:size],
                              port.buffer_discounts(buffers['terminations'][:size] != 0, 0.99))
    assert np.isnan(buffers['rewards'][size:]).all()
    infos = port_.update(buffers, workers, g['indices'], g['eps'])
    np.testing.assert_allclose([i['critic']['loss'] for i in infos], g['info/critic/loss'], rtol=1e-6)
    if kind != 'd4pg':
        q_key = 'info/critic/q_mean' if kind in ('ddpg', 'mpo') else 'info/critic/q1_mean'
        np.testing.assert_allclose([i['critic']['q1'] for i in infos], g[q_key], rtol=1e-5, atol=1e-7)
    if kind == 'mpo':
        for key in ('policy_mean_loss', 'policy_std_loss', 'kl_mean_loss', 'kl_std_loss',
                    'alpha_mean_loss', 'alpha_std_loss', 'alpha_mean', 'alpha_std'):
            np.testing.assert_allclose([i['actor'][key] for i in infos], g['info/actor/' + key],
                                       rtol=1e-5, atol=1e-7, err_msg=key)
        for key in ('temperature_loss', 'temperature', 'penalty_temperature'):
            np.testing.assert_allclose([i['actor'][key] for i in infos],
                                       g['info/actor/' + key + '_mean'], rtol=1e-5, err_msg=key)
    else:
        np.testing.assert_allclose([i['actor']['loss'] for i in infos if 'actor' in i],
                                   g['info/actor/loss'], rtol=1e-5, atol=1e-7)
    for key, value in port_.state().items():
        np.testing.assert_allclose(value, g['post/' + key], rtol=0, atol=1e-7, err_msg=key)


def test_ppo_minibatch_update_matches_reference(golden):
    """Segment(batch_size=64): shuffled minibatches (segments.py:58-65), ragged last batch, KL stop
    in the middle of an iteration."""
    g = golden('ppo_minibatch_small')
    seed, iterations, bs = int(g['cfg'][4]), int(g['cfg'][5]), int(g['batch_size'])
    actor, critic, norm = _params(g, 'pre0/')
    seg = {k: g[f'u0/segment/{k}'] for k in (
        'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
        'log_probs')}
    new_actor, new_critic, infos, _ = port.ppo_update(
        actor, critic, norm, seg, batch_iterations=iterations, batch_size=bs,
        np_random=np.random.RandomState(seed))
    assert len(infos) == int(g['u0/info/critic/iterations'][0])
    kl = np.array([i['actor']['kl'] for i in infos if 'actor' in i])
    np.testing.assert_allclose(kl, g['u0/info/actor/kl'], rtol=1e-5, atol=1e-5)
    assert np.array_equal([i['actor']['stop'] for i in infos if 'actor' in i], g['u0/info/actor/stop'])
    np.testing.assert_allclose([i['critic']['loss'] for i in infos], g['u0/info/critic/loss'],
                               rtol=1e-5, atol=1e-5)
    ref_actor, ref_critic, _ = _params(g, 'post0/')
    for got, want, before in zip(new_actor + new_critic, ref_actor + ref_critic, actor + critic):
        np.testing.assert_allclose(got - before, want - before, atol=2e-5, rtol=0)
