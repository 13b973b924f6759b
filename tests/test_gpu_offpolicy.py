"""GPU parity tests of the off-policy path (SAC / TD3): gemm16 building block, HBM Buffer
store / gather (bit-exact), policy forward, and whole learner updates against the reference
goldens (oracle/make_golden.py run_offpolicy) — through the C ABI / the drop-in agents."""
import numpy as np
import pytest

import numpy_port as port

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from tonic_amd import _lib
    assert torch.cuda.is_available()
    return _lib.load()


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda().contiguous()


@pytest.mark.parametrize('mode', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(1024, 256, 256), (100, 256, 119), (24, 1, 32), (37, 21, 88),
                                   (16, 32, 16), (1, 256, 1024), (256, 119, 1024), (256, 256, 100),
                                   (21, 256, 100)])
def test_gemm16_vs_numpy(lib, mode, M, N, K):
    from tonic_amd import _lib
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    a_shape = (M, K) if mode[0] == 'N' else (K, M)
    b_shape = (N, K) if mode[1] == 'T' else (K, N)
    a = rng.normal(size=a_shape).astype(np.float32)
    b = rng.normal(size=b_shape).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    mask = rng.normal(size=(M, N)).astype(np.float32)
    c0 = rng.normal(size=(M, N)).astype(np.float32)
    al = a if mode[0] == 'N' else a.T
    bl = b.T if mode[1] == 'T' else b
    ref = al.astype(np.float64) @ bl.astype(np.float64)
    tol = 2e-6 * np.sqrt(K) * max(1.0, np.abs(ref).max())
    da, db, dbias, dmask = dev(a), dev(b), dev(bias), dev(mask)
    out = torch.zeros(M, N).cuda()
    colsum = torch.zeros(M).cuda()
    _lib.check(lib.tonic_gemm_f32(mode.encode(), da.data_ptr(), db.data_ptr(), out.data_ptr(),
                                  None, None, colsum.data_ptr() if mode == 'TN' else None,
                                  M, N, K, a_shape[1], b_shape[1], N, 0, 0, 1.0, None), 'gemm')
    assert np.abs(out.cpu().numpy() - ref).max() <= tol
    if mode == 'TN':
        np.testing.assert_allclose(colsum.cpu().numpy(), a.sum(0), rtol=1e-5, atol=1e-4)
    # fused epilogue: bias + relu, relu-mask, accumulate, alpha
    out = dev(c0)
    _lib.check(lib.tonic_gemm_f32(mode.encode(), da.data_ptr(), db.data_ptr(), out.data_ptr(),
                                  dbias.data_ptr(), dmask.data_ptr(), None, M, N, K, a_shape[1],
                                  b_shape[1], N, 1, 1, 0.5, None), 'gemm-epilogue')
    want = c0 + np.where(mask > 0, np.maximum(0.5 * ref + bias, 0), 0)
    assert np.abs(out.cpu().numpy() - want).max() <= tol + 1e-6


def test_buffer_n_step_returns_bit_exact(lib, golden):
    """Buffer(return_steps > 1) on the GPU == the reference's accumulate_n_steps (buffers.py:58-79),
    bit for bit, incl. the early steps and the circular wrap."""
    import tonic_amd
    from test_oracle_golden import nstep_cases
    g = golden('buffer_nstep')
    for pre, workers, size, steps, index, filled, rows in nstep_cases(g):
        buf = tonic_amd.replays.Buffer(size=size, return_steps=steps)
        buf.initialize(seed=0)
        for t, row in enumerate(rows):
            buf.store(**{k: dev(v) for k, v in row.items()})
            if t == 3:
                for k, v in buf.buffers.items():
                    assert np.array_equal(v.cpu().numpy(), g[pre + 'early_' + k], equal_nan=True), k
        assert (buf.index, buf.size) == (index, filled)
        for k, v in buf.buffers.items():
            assert np.array_equal(v.cpu().numpy(), g[pre + 'buf_' + k], equal_nan=True), (pre, k)


@pytest.mark.parametrize('name', ['sac_small', 'td3_small'])
def test_buffer_store_gather_bit_exact(lib, golden, name):
    """Replays the reference run's stores into the HBM Buffer and gathers with the reference's
    index stream: contents, discounts, NaN padding and batches are bit-identical."""
    from tonic_amd.replays import Buffer
    g = golden(name)
    O, A, W, hidden, B, iterations, seed, loop_steps = (int(x) for x in g['cfg'])
    size = int(g['buffer_size'])
    buf = Buffer(size=400, batch_iterations=iterations, batch_size=B, steps_before_batches=W * 10,
                 steps_between_batches=W * 10)
    buf.initialize(seed=seed, device='cuda')
    ref = {k[len('buffer/'):]: g[k] for k in g.files if k.startswith('buffer/')}
    for t in range(size):
        buf.store(**{k: dev(ref[k][t]) for k in ('observations', 'actions', 'next_observations',
                                                 'rewards', 'resets', 'terminations')})
    assert buf.size == size and buf.max_size == ref['rewards'].shape[0]
    for k, want in ref.items():
        assert np.array_equal(buf.buffers[k].cpu().numpy(), want, equal_nan=True), k
    indices = buf.sample_indices()
    assert np.array_equal(indices, g['indices']), 'index stream must be bit-exact'
    for it in range(iterations):
        batch = buf.gather(dev(indices[it], torch.int64))
        rows, cols = indices[it] // W, indices[it] % W
        for k in ('observations', 'actions', 'next_observations', 'rewards', 'discounts'):
            assert np.array_equal(batch[k].cpu().numpy(), ref[k][rows, cols]), k


@pytest.mark.parametrize('rows,W,O,A,B', [(1000000, 1, 111, 8, 1024),      # BASELINE cfg 3
                                          (1953, 512, 67, 21, 100),         # BASELINE cfg 4 (reference batch)
                                          (1953, 512, 67, 21, 1024)])
def test_buffer_gather_bit_exact_at_baseline_size(lib, rows, W, O, A, B):
    """BASELINE's own buffers — cfg 3: 1 000 000 x 1 transitions of Ant-v3 shapes (0.94 GB), cfg 4:
    [1953, 512] of humanoid-walk shapes — full, and the 50 batches of one `Buffer.get`
    (buffers.py:81-91): the index stream is `RandomState(0).randint(size * W, size=B)` bit for bit,
    `rows = idx // W`, `cols = idx % W` (int64), and every gathered field equals NumPy's fancy
    indexing of the same arrays bit for bit — through the one-launch `gather_many` the agents use
    and through the per-batch `gather` / the drop-in `get`."""
    from tonic_amd.replays import Buffer
    buf = Buffer(size=rows * W, batch_size=B)
    buf.initialize(seed=0, device='cuda')
    buf._allocate(W, O, A)
    assert buf.max_size == rows
    fill = np.random.default_rng(rows + B)
    host = {}
    for key, tensor in buf.buffers.items():
        host[key] = fill.standard_normal(tuple(tensor.shape), dtype=np.float32)
        tensor.copy_(torch.from_numpy(host[key]))
    buf.index, buf.size = 0, rows
    stream = np.random.RandomState(0)
    want_indices = np.stack([stream.randint(rows * W, size=B) for _ in range(buf.batch_iterations)])
    indices = buf.sample_indices()
    assert indices.dtype == np.int64 and np.array_equal(indices, want_indices)
    assert indices.max() > rows * W * 0.99 and indices.min() < rows * W * 0.01   # the whole range
    r, c = want_indices // W, want_indices % W
    many = buf.gather_many(torch.as_tensor(indices, device='cuda'))
    for key in ('observations', 'actions', 'next_observations', 'rewards', 'discounts'):
        assert np.array_equal(many[key].cpu().numpy(), host[key][r, c]), key
    one = buf.gather(torch.as_tensor(indices[7], device='cuda'))
    for key in ('observations', 'actions', 'next_observations', 'rewards', 'discounts'):
        assert np.array_equal(one[key].cpu().numpy(), host[key][r[7], c[7]]), key
    # the drop-in generator draws the NEXT 50 batches of the same stream
    for it, batch in enumerate(buf.get('observations', 'rewards', steps=0)):
        idx = stream.randint(rows * W, size=B)
        assert np.array_equal(batch['observations'].cpu().numpy(), host['observations'][idx // W, idx % W])
        assert np.array_equal(batch['rewards'].cpu().numpy(), host['rewards'][idx // W, idx % W])
    assert it == buf.batch_iterations - 1


def _agent_from_golden(g, kind):
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    O, A, W, hidden, B, iterations, seed, loop_steps = (int(x) for x in g['cfg'])
    relu = torch.nn.ReLU
    sizes = (hidden, hidden)
    if 'torso_sizes' in g.files:
        sizes = tuple(int(x) for x in g['torso_sizes'])
        relu = getattr(torch.nn, str(g['torso_activation']))
    critic_head = (tt.models.DistributionalValueHead(*[int(v) if i == 2 else float(v)
                                                        for i, v in enumerate(g['atoms'])])
                   if kind == 'd4pg' else tt.models.ValueHead())
    critic = tt.models.Critic(encoder=tt.models.ObservationActionEncoder(),
                              torso=tt.models.MLP(sizes, relu), head=critic_head)
    if kind == 'sac':
        head = tt.models.GaussianPolicyHead(loc_activation=torch.nn.Identity,
                                            distribution=tt.models.SquashedMultivariateNormalDiag)
    elif kind == 'mpo':
        head = tt.models.GaussianPolicyHead()
    else:
        head = tt.models.DeterministicPolicyHead()
    container = (tt.models.ActorCriticWithTargets if kind in ('ddpg', 'd4pg', 'mpo')
                 else tt.models.ActorTwinCriticWithTargets)
    model = container(
        actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(),
                              torso=tt.models.MLP(sizes, relu), head=head),
        critic=critic, observation_normalizer=tt.normalizers.MeanStd())
    replay = tonic_amd.replays.Buffer(size=400, batch_iterations=iterations, batch_size=B,
                                      steps_before_batches=W * 10, steps_between_batches=W * 10,
                                      return_steps=int(g['return_steps']) if 'return_steps' in g.files else 1)
    if kind == 'sac':
        agent = tt.agents.SAC(model=model, replay=replay,
                              exploration=tonic_amd.explorations.NoActionNoise(start_steps=W * 5))
    elif kind == 'mpo':
        samples = int(g['samples'])
        agent = tt.agents.MPO(
            model=model, replay=replay,
            actor_updater=tt.updaters.MaximumAPosterioriPolicyOptimization(num_samples=samples),
            critic_updater=tt.updaters.ExpectedSARSA(num_samples=samples))
    else:
        cls = {'ddpg': tt.agents.DDPG, 'd4pg': tt.agents.D4PG, 'td3': tt.agents.TD3}[kind]
        agent = cls(model=model, replay=replay,
                    exploration=tonic_amd.explorations.NormalActionNoise(start_steps=W * 5))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    return agent


# (the last two: torsos outside the fused kernels' shape — unequal widths (100, 60), ELU (48, 40) — which run
#  layer by layer on gemm16 launches: `H` = tonic_mlp_hidden(H1, H2, activation))
GENERIC_TORSOS = ('sac_uneven_small', 'td3_elu_small')
OFFPOLICY_CASES = [('sac_small', 'sac'), ('td3_small', 'td3'), ('ddpg_small', 'ddpg'),
                   ('d4pg_small', 'd4pg'), ('mpo_small', 'mpo'), ('sac_uneven_small', 'sac'),
                   ('td3_elu_small', 'td3')]


@pytest.mark.parametrize('name,kind', [('sac_uneven_small', 'sac'), ('td3_elu_small', 'td3')])
def test_offpolicy_generic_torsos_as_stock_torch_operators(lib, golden, name, kind, monkeypatch):
    """What every torso outside the HIP paths runs on (three layers, other activations, ...): stock torch
    operators on the device — forced here for the two generic goldens with TONIC_AMD_TORSO_STOCK=1."""
    monkeypatch.setenv('TONIC_AMD_TORSO_STOCK', '1')
    test_offpolicy_update_matches_reference(lib, golden, name, kind, stock=True)


@pytest.mark.parametrize('name,kind', OFFPOLICY_CASES)
def test_offpolicy_update_matches_reference(lib, golden, name, kind, stock=False):
    g = golden(name)
    import tonic_amd.torch as tt
    agent = _agent_from_golden(g, kind)
    if name in GENERIC_TORSOS:
        for updater in (agent.critic_updater, agent.actor_updater):
            assert updater.stock == stock and (stock or updater.hidden >= 1024), (updater.stock, updater.hidden)
    state = agent.model.state_dict()
    for key in state:       # identical initialisation from the same seed (CPU init parity)
        np.testing.assert_array_equal(state[key].cpu().numpy(), g['init/' + key], err_msg=key)
    agent.model.load_state_dict({k[len('pre/'):]: torch.as_tensor(g[k]) for k in g.files
                                 if k.startswith('pre/')})
    before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    ref = {k[len('buffer/'):]: g[k] for k in g.files if k.startswith('buffer/')}
    for t in range(int(g['buffer_size'])):
        agent.replay.store(**{k: dev(ref[k][t]) for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations')})
    if agent.replay.return_steps > 1:
        # the captured buffer already holds the accumulated n-step rows (their bit-exactness is
        # test_buffer_n_step_returns_bit_exact's subject): install them as they are
        for k, v in ref.items():
            agent.replay.buffers[k].copy_(dev(v))
    infos = agent.enqueue_update(g['indices'], g['eps']).cpu().numpy()
    np.testing.assert_allclose(infos[0][:, 0], g['info/critic/loss'], rtol=1e-5, atol=1e-5)
    if kind in ('ddpg', 'mpo'):
        np.testing.assert_allclose(infos[0][:, 1], g['info/critic/q_mean'], rtol=1e-5, atol=1e-5)
    elif kind == 'd4pg':
        pass                # DistributionalDeterministicQLearning logs the loss only (critics.py:122)
    else:
        np.testing.assert_allclose(infos[0][:, 1], g['info/critic/q1_mean'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(infos[0][:, 2], g['info/critic/q2_mean'], rtol=1e-5, atol=1e-5)
    if kind == 'mpo':            # the eleven logged values of actors.py:449-464, every iteration
        stats = agent._mpo_stats.cpu().numpy()
        A = agent.action_size
        for i, key in enumerate(tt.updaters.MPO_INFO):
            suffix = '_mean' if key.startswith('temperature') else ''
            np.testing.assert_allclose(stats[:, i], g['info/actor/' + key + suffix], rtol=2e-5,
                                       atol=2e-6, err_msg=key)
        np.testing.assert_allclose(stats[:, 8:8 + A], g['info/actor/alpha_mean'], rtol=1e-5)
        np.testing.assert_allclose(stats[:, 8 + A:8 + 2 * A], g['info/actor/alpha_std'], rtol=1e-5)
        np.testing.assert_allclose(stats[:, 8 + 2 * A], g['info/actor/penalty_temperature_mean'],
                                   rtol=1e-5)
    else:
        ran = infos[1][:, 6] > 0
        assert ran.sum() == len(g['info/actor/loss'])
        np.testing.assert_allclose(infos[1][ran, 0], g['info/actor/loss'], rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    for key, start in before.items():
        if 'normalizer' in key:
            continue
        got = after[key].detach().cpu().numpy() - start
        want = g['post/' + key] - start
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5, err_msg=key)


@pytest.mark.parametrize('name,kind', OFFPOLICY_CASES)
def test_offpolicy_agent_drop_in_trajectory(lib, golden, name, kind):
    """agent.step / agent.update with NumPy in/out replays the reference run: warm-up actions
    from the NumPy stream, policy actions (+ exploration noise / sampled noise) after it, the
    learner update triggered at the same step, same statistics."""
    import tonic_amd
    g = golden(name)
    O, A, W, hidden, B, iterations, seed, loop_steps = (int(x) for x in g['cfg'])
    agent = _agent_from_golden(g, kind)
    env = tonic_amd.environments.distribute(
        lambda: tonic_amd.environments.Synthetic(O, A, max_episode_steps=5), 1, W)
    env.initialize(seed=seed)
    observations = env.start()
    rng = np.random.RandomState(seed + 1)
    updated = False
    for t in range(loop_steps):
        np.testing.assert_array_equal(observations, g['act/observations'][t])
        actions = agent.step(observations, t * W)
        if not updated:          # after the first update parameters differ at rounding level
            np.testing.assert_allclose(actions, g['act/actions'][t], rtol=0, atol=5e-6)
        observations, infos = env.step(g['act/actions'][t])
        infos['rewards'] = (infos['rewards'] + rng.normal(size=W)).astype(np.float32)
        term = rng.uniform(size=W) < 0.1
        infos['terminations'] = term
        infos['resets'] = infos['resets'] | term
        agent.update(**infos, steps=t * W)
        if hasattr(agent, 'last_infos') and not updated:
            updated = True
            infos_dev = agent.last_infos
            np.testing.assert_allclose(infos_dev[0][:, 0], g['info/critic/loss'], rtol=1e-4, atol=1e-5)
    assert updated


@pytest.mark.parametrize('kind,O,A,W,B,support', [
    ('sac', 111, 8, 1, 1024, None), ('td3', 67, 21, 64, 100, None), ('ddpg', 17, 6, 4, 100, None),
    ('d4pg', 24, 6, 4, 256, (-150., 150., 51)), ('d4pg', 6, 3, 4, 100, (-2., 12., 51)),
    ('mpo', 24, 6, 4, 100, None)])
def test_offpolicy_full_size_iteration_vs_oracle(lib, kind, O, A, W, B, support):
    """cfg-3 (SAC, O=111, A=8, B=1024) and cfg-4 per-GPU (TD3, O=67, A=21, 64 workers, the
    reference's default B=100) shapes with the default 256-wide networks: two learner iterations
    on the HIP path vs the torch-CPU oracle from identical parameters, buffer, indices, noise."""
    import tonic_amd
    import tonic_amd.torch as tt
    import torch_port
    from tonic_amd.environments import Box
    torch.set_num_threads(8)
    rng = np.random.RandomState(7)
    rows = 64
    replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=2, batch_size=B)
    agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3, ddpg=tt.agents.DDPG,
                 d4pg=tt.agents.D4PG, mpo=tt.agents.MPO)[kind](replay=replay)    # (d4pg: 51 atoms on +-150)
    if support is not None:         # (the target networks are copies made at construction)
        agent.model.critic.head = tt.models.DistributionalValueHead(*support)
        agent.model.target_critic.head = tt.models.DistributionalValueHead(*support)
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
    # make the normaliser non-trivial
    norm = agent.model.observation_normalizer
    norm._mean.data.copy_(dev(rng.normal(size=O) * 0.3))
    norm._std.data.copy_(dev(np.abs(rng.normal(size=O)) + 0.5))
    state = {'pre/' + k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)),
                rewards=rng.normal(size=(rows, W)) * (0.4 * support[1] if kind == 'd4pg' else 1.0),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    for t in range(rows):
        replay.store(**{k: dev(v[t]) for k, v in host.items()})
    host['discounts'] = port.buffer_discounts(host['terminations'] != 0, 0.99)
    indices = replay.sample_indices()
    draws = 2 if kind in ('sac', 'mpo') else 1
    eps = rng.normal(size=(2, draws, B * (20 if kind == 'mpo' else 1), A)).astype(np.float32)
    oracle = torch_port.OffPolicyPort(kind, state, 'pre/', atoms=support or (-150., 150., 51))
    want = oracle.update(host, W, indices, eps)
    infos = agent.enqueue_update(indices, eps).cpu().numpy()
    np.testing.assert_allclose(infos[0][:, 0], [i['critic']['loss'] for i in want], rtol=1e-5, atol=1e-5)
    if kind != 'd4pg':
        np.testing.assert_allclose(infos[0][:, 1], [i['critic']['q1'] for i in want], rtol=1e-5, atol=1e-5)
    if kind == 'mpo':
        stats = agent._mpo_stats.cpu().numpy()
        for i, key in enumerate(tt.updaters.MPO_INFO):
            np.testing.assert_allclose(stats[:, i], [w['actor'][key] for w in want], rtol=2e-5,
                                       atol=2e-6, err_msg=key)
        np.testing.assert_allclose(agent.actor_updater.duals.cpu().numpy(),
                                   np.concatenate([d.detach().numpy() for d in oracle.duals]),
                                   rtol=0, atol=1e-6)
    else:
        ran = infos[1][:, 6] > 0
        np.testing.assert_allclose(infos[1][ran, 0],
                                   [i['actor']['loss'] for i in want if 'actor' in i],
                                   rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    for key, value in oracle.state().items():
        got = after[key].detach().cpu().numpy() - state['pre/' + key]
        np.testing.assert_allclose(got, value - state['pre/' + key], rtol=0, atol=1e-5, err_msg=key)


@pytest.mark.parametrize('kind,O,A,hidden,B', [
    ('sac', 17, 6, 1024, 64), ('td3', 17, 6, 1024, 64),
    # the corners of the image passes (csrc/mlpimg_body.h): inputs wider than 256 columns (two passes over the
    # row, Humanoid-v3's 376 + 17), widths of 16 mod 32 (the zeroed upper half of the last k-chunk), more than 32
    # actions (two k-chunks per head in the actor's chain), batches that are no multiple of 16
    ('sac', 376, 17, 48, 40), ('td3', 300, 6, 80, 33), ('sac', 60, 30, 256, 50), ('td3', 9, 33, 112, 100)])
def test_plain_torsos_of_other_widths_vs_oracle(lib, kind, O, A, hidden, B):
    """Equal-width ReLU torsos other than the default 256 on the HIP entries — never on stock torch operators:
    `updater.stock is False`.  1 024 units (a PLAIN width of the C ABI's `H` argument, tonic_mlp_hidden
    packs only unequal / non-ReLU torsos, bit 30 set) run layer by layer (csrc/gemm16.hip); widths up to 256 on
    the fused passes and their fp16x2 weight images, here at the shapes' corners.  Two learner iterations agree with
    the torch-CPU oracle from identical parameters, buffer, indices and noise."""
    import tonic_amd
    import tonic_amd.torch as tt
    import torch_port
    from tonic_amd.environments import Box
    assert lib.tonic_mlp_hidden(1024, 1024, 1) == 1024 and lib.tonic_mlp_hidden(400, 300, 1) & (1 << 30)
    W, rows = 2, 48
    rng = np.random.RandomState(17)
    relu = torch.nn.ReLU
    head = (tt.models.GaussianPolicyHead(loc_activation=torch.nn.Identity,
                                         distribution=tt.models.SquashedMultivariateNormalDiag)
            if kind == 'sac' else tt.models.DeterministicPolicyHead())
    model = tt.models.ActorTwinCriticWithTargets(
        actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(),
                              torso=tt.models.MLP((hidden, hidden), relu), head=head),
        critic=tt.models.Critic(encoder=tt.models.ObservationActionEncoder(),
                                torso=tt.models.MLP((hidden, hidden), relu), head=tt.models.ValueHead()),
        observation_normalizer=tt.normalizers.MeanStd())
    replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=2, batch_size=B)
    agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3)[kind](model=model, replay=replay)
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
    assert agent.critic_updater.stock is False and agent.actor_updater.stock is False
    assert agent.hidden == hidden and (agent._fused_kind() is not None) == (hidden <= 256)
    state = {'pre/' + k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)), rewards=rng.normal(size=(rows, W)),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    for t in range(rows):
        replay.store(**{k: dev(v[t]) for k, v in host.items()})
    host['discounts'] = port.buffer_discounts(host['terminations'] != 0, 0.99)
    indices = replay.sample_indices()
    eps = rng.normal(size=(2, 2 if kind == 'sac' else 1, B, A)).astype(np.float32)
    oracle = torch_port.OffPolicyPort(kind, state, 'pre/')
    want = oracle.update(host, W, indices, eps)
    infos = agent.enqueue_update(indices, eps).cpu().numpy()
    np.testing.assert_allclose(infos[0][:, 0], [i['critic']['loss'] for i in want], rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    lr = agent.actor_updater.hyper['lr']
    for key, value in oracle.state().items():
        got = after[key].detach().cpu().numpy() - state['pre/' + key]
        diff = np.abs(got - (value - state['pre/' + key]))
        # Adam's first steps move every element by ~lr whatever its gradient's size: an element whose gradient is at
        # float32 rounding level takes a step of its own (seen: 1 of 1 048 576 in actor.torso.model.2.weight at 1 024
        # units, 1.6e-5; 1 of 12 544 in critic_1.torso.model.2.weight at 112 units, 6.2e-5); the reference itself does
        # that when its summation order changes.  At most one such element per tensor (1e-5 of a large one)
        violations = int((diff > 1e-5).sum())
        assert violations <= max(1, int(1e-5 * diff.size)) and diff.max() <= 2 * lr, (key, diff.max(), violations)


@pytest.mark.parametrize('kind,O,A,W,B,hidden', [
    ('sac', 111, 8, 1, 1024, 256), ('td3', 67, 21, 64, 100, 256), ('ddpg', 17, 6, 4, 100, 256),
    ('sac', 11, 3, 4, 24, 32), ('td3', 9, 4, 3, 37, 48), ('sac', 40, 30, 2, 50, 256),
    # (more workgroups than the chip holds at once: the chained launches' waits must follow the
    #  dispatch order — 313 tiles x 4 roles)
    ('sac', 17, 6, 8, 5000, 256), ('td3', 17, 6, 8, 5000, 256)])
def test_fused_iteration_equals_the_split_entry_points(lib, monkeypatch, kind, O, A, W, B, hidden):
    """tonic_q_iteration (both policy passes in one launch, head backward folded into the actor's
    backward launch, Adam + polyak in the epilogue of the weight-gradient launches, all batches
    gathered up front) against tonic_twin_q_grad + tonic_adam_step + tonic_actor_q_grad +
    tonic_adam_polyak_step: the same expressions in the same order, so online and target
    parameters, Adam moments, step counters and logged statistics must agree BIT FOR BIT after six
    iterations (TD3: three actor steps) — eager and replayed from a hipGraph."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    rng = np.random.RandomState(11)
    rows, iterations = 48, 6
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)), rewards=rng.normal(size=(rows, W)),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    draws = 2 if kind == 'sac' else 1
    eps = rng.normal(size=(iterations, draws, B, A)).astype(np.float32)
    indices = rng.randint(rows * W, size=(iterations, B))
    results = {}
    for mode, env, graph in (('split', '0', False), ('fused', '1', False), ('fused-graph', '1', True)):
        monkeypatch.setenv('TONIC_AMD_FUSED_ITERATION', env)
        relu = torch.nn.ReLU
        head = (tt.models.GaussianPolicyHead(loc_activation=torch.nn.Identity,
                                             distribution=tt.models.SquashedMultivariateNormalDiag)
                if kind == 'sac' else tt.models.DeterministicPolicyHead())
        container = (tt.models.ActorCriticWithTargets if kind == 'ddpg'
                     else tt.models.ActorTwinCriticWithTargets)
        model = container(
            actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(),
                                  torso=tt.models.MLP((hidden, hidden), relu), head=head),
            critic=tt.models.Critic(encoder=tt.models.ObservationActionEncoder(),
                                    torso=tt.models.MLP((hidden, hidden), relu),
                                    head=tt.models.ValueHead()),
            observation_normalizer=tt.normalizers.MeanStd())
        replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations, batch_size=B)
        agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3, ddpg=tt.agents.DDPG)[kind](
            model=model, replay=replay)
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
        assert (agent._fused_kind() is not None) == (mode != 'split')
        norm = agent.model.observation_normalizer
        norm._mean.data.copy_(dev(np.random.RandomState(1).normal(size=O) * 0.3))
        norm._std.data.copy_(dev(np.abs(np.random.RandomState(2).normal(size=O)) + 0.5))
        for t in range(rows):
            replay.store(**{k: dev(v[t]) for k, v in host.items()})
        infos = [agent.enqueue_update(indices, eps, graph=graph).cpu().numpy().copy()
                 for _ in range(2)]                       # twice: the second call replays the graph
        results[mode] = dict(
            infos0=infos[0], infos1=infos[1], online=agent.model.flat_online.cpu().numpy(),
            target=agent.model.flat_target.cpu().numpy(),
            critic_m=agent.critic_updater.exp_avg.cpu().numpy(),
            critic_v=agent.critic_updater.exp_avg_sq.cpu().numpy(),
            actor_m=agent.actor_updater.exp_avg.cpu().numpy(),
            actor_v=agent.actor_updater.exp_avg_sq.cpu().numpy(),
            steps=np.array([int(agent.critic_updater.state[0]), int(agent.actor_updater.state[0])]))
    due = iterations if kind != 'td3' else iterations // 2
    assert list(results['split']['steps']) == [2 * iterations, 2 * due]
    assert np.abs(results['split']['online']).max() > 0 and np.isfinite(results['split']['online']).all()
    for mode in ('fused', 'fused-graph'):
        for key, want in results['split'].items():
            got = results[mode][key]
            if not np.array_equal(got, want):           # (say WHERE: NaN counts as a mismatch)
                where = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
                first = tuple(where[0]) if len(where) else ()
                raise AssertionError((mode, key, len(where), first,
                                      got[first] if first else None, want[first] if first else None))


@pytest.mark.parametrize('kind', ['sac', 'td3'])
def test_update_call_in_chunks_is_bit_identical(lib, monkeypatch, kind):
    """DDPG._update draws the index and noise streams of an update call chunk by chunk (10 iterations per captured
    graph, two alternating sets of graph inputs) so that the host's draws run under the GPU's work: the same draws in
    the same order — parameters, targets, moments, step counters and every logged statistic of two update calls of
    40 iterations are bit-identical with the one-graph call (TONIC_AMD_UPDATE_CHUNK=0), and the generators end up
    in the same state."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    O, A, W, B, rows, iterations = (23, 5, 2, 64, 40, 40)
    rng = np.random.RandomState(21)
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)), rewards=rng.normal(size=(rows, W)),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    results = {}
    for mode, chunk in (('one graph', '0'), ('chunks', '10')):
        monkeypatch.setenv('TONIC_AMD_UPDATE_CHUNK', chunk)
        replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations, batch_size=B)
        agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3)[kind](replay=replay)
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
        assert (agent._update_chunk(iterations) == 10) == (mode == 'chunks')
        norm = agent.model.observation_normalizer
        for t in range(rows):
            replay.store(normalizer=norm, **{k: dev(v[t]) for k, v in host.items()})
        infos = []
        for call in range(2):
            agent._update(steps=100000 + 50 * call)
            infos.append(agent.last_infos.copy())
            for t in range(3):               # (the normaliser's update wants new records every time)
                replay.store(normalizer=norm, **{k: dev(v[t]) for k, v in host.items()})
        results[mode] = dict(
            infos0=infos[0], infos1=infos[1], online=agent.model.flat_online.cpu().numpy(),
            target=agent.model.flat_target.cpu().numpy(),
            critic_m=agent.critic_updater.exp_avg.cpu().numpy(), actor_v=agent.actor_updater.exp_avg_sq.cpu().numpy(),
            steps=np.array([int(agent.critic_updater.state[0]), int(agent.actor_updater.state[0])]),
            next_index=replay.np_random.randint(1 << 30, size=4), next_noise=torch.randn(4).numpy())
    assert list(results['chunks']['steps']) == [2 * iterations, 2 * iterations // (2 if kind == 'td3' else 1)]
    for key, want in results['one graph'].items():
        got = results['chunks'][key]
        assert np.array_equal(got, want, equal_nan=True), (key, np.abs(got - want).max())


def test_capture_holds_the_garbage_collector_off(lib):
    """A cyclic collection that starts inside a stream capture may finalize an OLDER captured graph that earlier
    work left behind in a reference cycle: hipGraphExecDestroy while a stream captures throws from ~CUDAGraph, i.e.
    terminates the process (scripts/capture_gc_probe.py; seen as `Fatal Python error: Aborted ... Garbage-collecting`
    in the middle of a test session).  `_lib.capturing` — every capture of the package — collects first and keeps
    the collector off until the capture has ended."""
    import gc
    import weakref
    from tonic_amd import _lib
    x = torch.zeros(64, device='cuda')

    class Holder:
        pass
    holder = Holder()
    holder.me = holder
    holder.graph = torch.cuda.CUDAGraph()
    with _lib.capturing(holder.graph):
        x.add_(1)
    holder.graph.replay()
    torch.cuda.synchronize()
    gone = weakref.ref(holder)
    del holder
    assert gc.isenabled()
    graph = torch.cuda.CUDAGraph()
    with _lib.capturing(graph):
        assert not gc.isenabled() and gone() is None       # collected BEFORE the capture began
        junk = [[i] for i in range(5000)]                  # (allocations that would have started a collection)
        x.add_(1)
    del junk
    assert gc.isenabled()
    graph.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 2.0                              # (two replays; a capture runs nothing)


@pytest.mark.parametrize('graph', [True, False])
@pytest.mark.parametrize('O,A,W,B,delay,rides', [(67, 21, 64, 100, 2, True), (23, 5, 2, 64, 3, True),
                                                 (111, 8, 1, 1024, 2, False)])
def test_policy_passes_ahead_are_bit_identical(lib, monkeypatch, graph, O, A, W, B, delay, rides):
    """TD3 steps its actor every `delay_steps` iterations (td3.py:43-46): the policy passes of iteration k + 1 read
    nothing a critic step writes, so they ride as more workgroups in the critic-step launch of an iteration k that
    leaves the actor alone (tonic_q_iteration_t.ahead: the workspace's other set of launch-1 outputs, exchange area
    and failure word; iteration k + 1 starts behind them) — when the launch still fits the chip at once
    (tonic_q_iteration_ahead_supported: not at B = 1 024).  The same kernels' code on the same inputs: parameters,
    targets, moments, step counters and every logged statistic of two update calls are bit-identical with every
    iteration by itself (TONIC_AMD_POLICY_AHEAD=0), replayed from a hipGraph and eager."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    rows, iterations = 40, 12
    rng = np.random.RandomState(33)
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)), rewards=rng.normal(size=(rows, W)),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    monkeypatch.setenv('TONIC_AMD_UPDATE_CHUNK', '0')
    assert bool(lib.tonic_q_iteration_ahead_supported(B, O, 256, A, 2, 2)) == rides
    results = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TONIC_AMD_POLICY_AHEAD', mode)
        replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations, batch_size=B)
        agent = tt.agents.TD3(replay=replay, delay_steps=delay)
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
        norm = agent.model.observation_normalizer
        for t in range(rows):
            replay.store(normalizer=norm, **{k: dev(v[t]) for k, v in host.items()})
        infos = []
        for call in range(2):
            indices = replay.sample_indices()
            eps = agent._draw_noise(indices.shape[0])
            infos.append(agent.enqueue_update(indices, eps, graph=graph).cpu().numpy().copy())
        results[mode] = dict(
            infos0=infos[0], infos1=infos[1], online=agent.model.flat_online.cpu().numpy(),
            target=agent.model.flat_target.cpu().numpy(),
            critic_m=agent.critic_updater.exp_avg.cpu().numpy(), actor_v=agent.actor_updater.exp_avg_sq.cpu().numpy(),
            steps=np.array([int(agent.critic_updater.state[0]), int(agent.actor_updater.state[0])]))
    assert list(results['1']['steps']) == [2 * iterations, 2 * iterations // delay]
    assert np.isfinite(results['1']['infos1'][0, :, 0]).all()
    for key, want in results['0'].items():
        got = results['1'][key]
        assert np.array_equal(got, want, equal_nan=True), (key, np.abs(got - want).max())


@pytest.mark.parametrize('kind,O,A,W,B', [('sac', 111, 8, 1, 1024), ('td3', 67, 21, 64, 100), ('ddpg', 17, 6, 4, 100)])
def test_fused_iteration_in_phases_equals_the_split_entry_points(lib, monkeypatch, kind, O, A, W, B):
    """Whenever something must see the complete gradient sums between gradients and step — here a gradient-norm
    clip on both updaters; the exchange between ranks in tests/test_gpu_multirank.py — the fused iteration runs in
    two halves (tonic_q_iteration_t.phase 1 / 2: the same chained launches, sums only) around the updaters' own
    clip + Adam [+ polyak] launches.  Bit for bit the split entry points' result (TONIC_AMD_FUSED_PHASES=0),
    eager and replayed from a hipGraph."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    rng = np.random.RandomState(12)
    rows, iterations = 48, 6
    host = dict(observations=rng.normal(size=(rows, W, O)), actions=rng.uniform(-1, 1, (rows, W, A)),
                next_observations=rng.normal(size=(rows, W, O)), rewards=rng.normal(size=(rows, W)),
                resets=rng.uniform(size=(rows, W)) < 0.1, terminations=rng.uniform(size=(rows, W)) < 0.05)
    host = {k: np.asarray(v, np.float32) for k, v in host.items()}
    eps = rng.normal(size=(iterations, 2 if kind == 'sac' else 1, B, A)).astype(np.float32)
    indices = rng.randint(rows * W, size=(iterations, B))
    results = {}
    for mode, env, graph in (('split', '0', False), ('phases', '1', False), ('phases-graph', '1', True)):
        monkeypatch.setenv('TONIC_AMD_FUSED_PHASES', env)
        updaters = dict(
            sac=(tt.updaters.TwinCriticSoftDeterministicPolicyGradient, tt.updaters.TwinCriticSoftQLearning),
            td3=(tt.updaters.DeterministicPolicyGradient, tt.updaters.TwinCriticDeterministicQLearning),
            ddpg=(tt.updaters.DeterministicPolicyGradient, tt.updaters.DeterministicQLearning))[kind]
        agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3, ddpg=tt.agents.DDPG)[kind](
            replay=tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations, batch_size=B),
            actor_updater=updaters[0](gradient_clip=0.7), critic_updater=updaters[1](gradient_clip=1.3))
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=5)
        assert (agent._fused_kind() is not None) == (mode != 'split')
        assert mode == 'split' or agent._fused_in_phases()
        for t in range(rows):
            agent.replay.store(**{k: dev(v[t]) for k, v in host.items()})
        infos = [agent.enqueue_update(indices, eps, graph=graph).cpu().numpy().copy() for _ in range(2)]
        results[mode] = dict(
            infos0=infos[0], infos1=infos[1], online=agent.model.flat_online.cpu().numpy(),
            target=agent.model.flat_target.cpu().numpy(),
            critic_m=agent.critic_updater.exp_avg.cpu().numpy(), actor_v=agent.actor_updater.exp_avg_sq.cpu().numpy(),
            steps=np.array([int(agent.critic_updater.state[0]), int(agent.actor_updater.state[0])]))
    assert np.abs(results['split']['online']).max() > 0 and np.isfinite(results['split']['online']).all()
    for mode in ('phases', 'phases-graph'):
        for key, want in results['split'].items():
            assert np.array_equal(results[mode][key], want), (mode, key)


def test_a_lost_workgroup_of_a_chained_launch_skips_the_step_and_raises(lib):
    """The workgroups of the chained launches wait for each other's values (exchange_read,
    csrc/mlpfwd.h).  A value that never comes — here: the first target workgroup of ONE critic
    step returns without a word (tuning key `chain_fault`) — must not hang the device and must not
    touch the model: its readers give up after 250 ms, the iteration's failure word makes BOTH
    optimizer epilogues of that iteration skip their step (no parameter, moment, target or step
    counter moves), the logged loss is NaN with the give-up mark in slot 7, the next iteration runs
    normally, and `agent._update` raises.  The device stays usable: a fresh agent reproduces the
    first update bit for bit."""
    import time
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd import _lib
    from tonic_amd.environments import Box
    O, A, W, B, rows = 17, 6, 4, 64, 48

    def make(iterations):
        rng = np.random.RandomState(3)
        replay = tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations, batch_size=B)
        agent = tt.agents.DDPG(replay=replay)           # (actor + targets every iteration)
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=1)
        assert agent._fused_kind() is not None
        for _ in range(rows):
            replay.store(observations=dev(rng.normal(size=(W, O)).astype(np.float32)),
                         actions=dev(rng.uniform(-1, 1, (W, A)).astype(np.float32)),
                         next_observations=dev(rng.normal(size=(W, O)).astype(np.float32)),
                         rewards=dev(rng.normal(size=W).astype(np.float32)),
                         resets=dev(np.zeros(W, np.float32)),
                         terminations=dev(np.zeros(W, np.float32)))
        return agent

    def snapshot(agent):
        m = agent.model
        tensors = [m.flat_online, m.flat_target, agent.critic_updater.exp_avg,
                   agent.critic_updater.exp_avg_sq, agent.actor_updater.exp_avg,
                   agent.actor_updater.exp_avg_sq, agent.critic_updater.state[:1],
                   agent.actor_updater.state[:1]]
        return [t.detach().cpu().clone() for t in tensors]

    rng = np.random.RandomState(4)
    eps = rng.normal(size=(2, 1, B, A)).astype(np.float32)
    indices = rng.randint(rows * W, size=(2, B))
    agent = make(2)
    good = agent.enqueue_update(indices, eps, graph=False).cpu().numpy().copy()
    assert np.isfinite(good[:, :, 0]).all() and not good[..., 7].any()
    after_good = snapshot(agent)
    # a fault in the FIRST of two iterations == only the second iteration ran
    reference = make(2)
    reference.enqueue_update(indices, eps, graph=False)
    torch.cuda.synchronize()
    # (the optimizer constants of a step follow the device's counter: iteration 2 alone is step 3)
    only_second = reference.enqueue_update(indices[1:], eps[1:], graph=False).cpu().numpy().copy()
    want = snapshot(reference)
    _lib.check(lib.tonic_set_tuning(b'chain_fault', 1), 'tuning')
    start = time.perf_counter()
    bad = agent.enqueue_update(indices, eps, graph=False).cpu().numpy().copy()
    elapsed = time.perf_counter() - start
    assert np.isnan(bad[0, 0, 0]) and np.isnan(bad[1, 0, 0]), bad[:, 0, 0]
    assert bad[0, 0, 7] == 1 and bad[1, 0, 7] == 1 and bad[0, 0, 6] == 0 and bad[1, 0, 6] == 0
    assert bad[0, 1, 7] == 0 and bad[1, 1, 7] == 0 and bad[0, 1, 6] == 1 and bad[1, 1, 6] == 1
    assert 0.2 < elapsed < 5.0, elapsed                 # the 250 ms bound, not a watchdog reset
    got = snapshot(agent)
    assert int(got[-2]) == int(after_good[-2]) + 1 and int(got[-1]) == int(after_good[-1]) + 1
    for g, w in zip(got[:2], want[:2]):                 # the model: as if iteration 1 had not been
        assert torch.isfinite(g).all()
    # the skipped iteration left the state of the first update; then ONE regular iteration ran.
    # Its Adam constants were formed for step 4 (the host counted the skipped step), the
    # reference's for step 3: compare the critic's logged row, which does not depend on them
    np.testing.assert_array_equal(bad[0, 1, :3], only_second[0, 0, :3])
    with pytest.raises(_lib.TonicHipError, match='gave up'):
        from tonic_amd.torch import agents
        agents._check_chain(bad)
    # the device and the library are fine: a fresh agent reproduces the first update bit for bit
    again = make(2).enqueue_update(indices, eps, graph=False).cpu().numpy()
    assert np.array_equal(again, good)


def test_adam_polyak_step_equals_the_two_calls(lib):
    """tonic_adam_polyak_step == tonic_adam_step on the block followed by tonic_polyak_update of
    the whole target buffer, bit for bit (block at the start, in the middle, at the end)."""
    from tonic_amd import _lib
    p = _lib.ptr
    total, coeff = 70_000, 0.005
    gen = torch.Generator(device='cuda').manual_seed(3)
    for offset, n in ((0, 20_003), (12_345, 33_333), (50_000, 20_000)):
        online = torch.randn(total, device='cuda', generator=gen)
        target = torch.randn(total, device='cuda', generator=gen)
        grads = torch.randn(n + 8, device='cuda', generator=gen)
        results = []
        for fused in (False, True):
            on, tg = online.clone(), target.clone()
            m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
            state = torch.zeros(4, dtype=torch.int32, device='cuda')
            info = torch.zeros(8, device='cuda')
            for _ in range(3):
                if fused:
                    _lib.check(lib.tonic_adam_polyak_step(
                        p(on), p(grads), p(m), p(v), p(state), offset, n, total, 1 / 64, 3e-4, 0.9,
                        0.999, 1e-8, 4, p(info), p(tg), coeff, None), 'fused')
                else:
                    block = on[offset:offset + n]
                    _lib.check(lib.tonic_adam_step(
                        p(block), p(grads), p(m), p(v), p(state), n, 1 / 64, 3e-4, 0.9, 0.999, 1e-8,
                        4, 0.0, 0.0, None, p(info), None, None), 'adam')
                    _lib.check(lib.tonic_polyak_update(p(tg), p(on), total, coeff, None), 'polyak')
            torch.cuda.synchronize()
            results.append((on.cpu(), tg.cpu(), m.cpu(), v.cpu(), state.cpu(), info.cpu()))
        for a, b in zip(*results):
            assert torch.equal(a, b)


@pytest.mark.parametrize('kind,o_dim,a_dim', [('sac', 111, 8), ('sac', 11, 3), ('td3', 67, 21),
                                              ('ddpg', 17, 6), ('sac', 9, 40), ('sac', 23, 20),
                                              ('td3', 5, 1)])
def test_policy_tail_in_the_forward_launch_is_bit_identical(lib, kind, o_dim, a_dim):
    """Sampling / target noise / dense copy folded into the actor's forward launch (tuning key
    policy_tail = 1, the default) against their own launches (0): same parameters, bit for bit,
    after whole learner iterations; same actions from tonic_policy_forward."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd import _lib
    from tonic_amd.environments import Box
    iters, rows, B = 3, 512, 96
    results, actions = [], []
    try:
        for tail in (1, 0):
            _lib.check(lib.tonic_set_tuning(b'policy_tail', tail), 'tuning')
            torch.manual_seed(0)
            np.random.seed(0)
            replay = tonic_amd.replays.Buffer(size=rows, batch_iterations=iters, batch_size=B)
            agent = {'sac': tt.agents.SAC, 'td3': tt.agents.TD3, 'ddpg': tt.agents.DDPG}[kind](
                replay=replay)
            agent.initialize(Box(-np.inf, np.inf, (o_dim,)), Box(-1, 1, (a_dim,)), seed=0)
            replay._allocate(1, o_dim, a_dim)
            gen = torch.Generator(device='cuda').manual_seed(1)
            for k, b in replay.buffers.items():
                b.copy_(torch.randn(b.shape, device='cuda', generator=gen) *
                        (0.0 if k in ('resets', 'terminations') else 1.0))
            replay.buffers['discounts'].fill_(0.99)
            replay.size = rows
            agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=False)
            torch.cuda.synchronize()
            results.append({k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()})
            obs = torch.randn(33, o_dim, device='cuda', generator=gen)
            out = torch.zeros(33, a_dim, device='cuda')
            eps = torch.randn(33, a_dim, device='cuda', generator=gen)
            ws = torch.empty(lib.tonic_offpolicy_workspace_bytes(33, o_dim, a_dim, 256),
                             dtype=torch.uint8, device='cuda')
            for with_eps in (True, False):
                _lib.check(lib.tonic_policy_forward(
                    _lib.ptr(agent.model.flat_actor.flat), _lib.ptr(obs),
                    _lib.ptr(eps) if with_eps else None, _lib.ptr(out), 1 if kind == 'sac' else 0,
                    33, o_dim, 256, a_dim, _lib.ptr(ws), ws.numel(), None), 'policy')
                actions.append(out.cpu().clone())
    finally:
        _lib.check(lib.tonic_set_tuning(b'policy_tail', 1), 'tuning')
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k
    assert torch.equal(actions[0], actions[2]) and torch.equal(actions[1], actions[3])
    assert all(torch.isfinite(v).all() for v in results[0].values())


@pytest.mark.parametrize('n,coeff', [(292114, 0.005), (1, 0.005), (70001, 0.25), (4097, 1.0)])
def test_polyak_update_bit_exact_vs_oracle(lib, n, coeff):
    """tonic_polyak_update against numpy_port.polyak (actor_critics.py:126-130: t.mul_(1-c);
    t.add_(c*o), three float32 roundings) bit for bit — three consecutive updates, values over
    many binades, and the polyak half of tonic_adam_polyak_step against the same oracle."""
    from tonic_amd import _lib
    rng = np.random.RandomState(n)
    scale = np.exp(rng.uniform(-20, 20, n)).astype(np.float32)
    online = (rng.standard_normal(n).astype(np.float32) * scale)
    target = (rng.standard_normal(n).astype(np.float32) * scale[::-1])
    d_online, d_target = dev(online), dev(target)
    want = target
    for _ in range(3):
        _lib.check(lib.tonic_polyak_update(d_target.data_ptr(), d_online.data_ptr(), n, coeff, None),
                   'tonic_polyak_update')
        want = port.polyak([want], [online], coeff)[0]
    torch.cuda.synchronize()
    assert np.array_equal(d_target.cpu().numpy(), want)
    # fused form: zero gradients leave the online block where it is, the targets still move
    d_target2 = dev(target)
    grads = torch.zeros(n + 8, device='cuda')
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    state = torch.zeros(4, dtype=torch.int32, device='cuda')
    info = torch.zeros(8, device='cuda')
    _lib.check(lib.tonic_adam_polyak_step(
        d_online.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(), 0, n, n,
        1.0, 3e-4, 0.9, 0.999, 1e-8, 0, info.data_ptr(), d_target2.data_ptr(), coeff, None), 'fused')
    torch.cuda.synchronize()
    assert np.array_equal(d_online.cpu().numpy(), online), 'a zero gradient must not move Adam'
    assert np.array_equal(d_target2.cpu().numpy(), port.polyak([target], [online], coeff)[0])
