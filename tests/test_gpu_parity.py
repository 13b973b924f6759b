"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle
(oracle/numpy_port.py) and the golden vectors produced by the unmodified reference.

Tolerances (north_star): index / integer work bit-exact; float32 losses and parameter
deltas within 1e-5.  Returns from the 1-chunk GAE scan and the MeanStd sums are required to
be BIT-exact because the kernels replay the reference's float32 operation order.
"""
import numpy as np
import pytest

import numpy_port as port
from test_oracle_golden import PPO_CASES, _params
from conftest import variant_library

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from tonic_amd import _lib
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return _lib.load()


def dev(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(device='cuda', dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda().contiguous()


def flat(params):
    return np.concatenate([np.asarray(p, np.float32).reshape(-1) for p in params])


def run_gae(lib, nv, rew, rst, term, val, gamma, lam, chunks):
    from tonic_amd import _lib
    T, W = rew.shape
    d = [dev(a) for a in (nv, rew, rst, term, val)]
    ret, adv = torch.empty(T, W).cuda(), torch.empty(T, W).cuda()
    stats = torch.zeros(4).cuda()
    ws = torch.empty(max(lib.tonic_gae_workspace_bytes(T, W, chunks), 16), dtype=torch.uint8).cuda()
    _lib.check(lib.tonic_gae_lambda_returns(
        *[t.data_ptr() for t in d], ret.data_ptr(), adv.data_ptr(), stats.data_ptr(), None, T, W,
        float(gamma), float(lam), chunks, ws.data_ptr(), ws.numel(), None), 'gae')
    torch.cuda.synchronize()
    return ret.cpu().numpy(), adv.cpu().numpy(), stats.cpu().numpy()


def normalise(adv, stats):
    return (adv - stats[0]) / stats[1] if stats[3] else adv


# ------------------------------------------------------------------------------- GAE

def test_gae_golden_bit_exact(lib, golden):
    g = golden('lambda_returns')
    for i in range(int(g['n_cases'])):
        for j in range(4):
            k = f'c{i}_{j}_'
            args = [g[k + n] for n in ('next_values', 'rewards', 'resets', 'terminations', 'values')]
            gamma, lam = float(g[k + 'gamma']), float(g[k + 'lambda'])
            ret, adv, stats = run_gae(lib, *args, gamma, lam, 1)
            assert np.array_equal(ret, g[k + 'returns']), f'{k}: 1-chunk scan must be bit-exact'
            np.testing.assert_allclose(normalise(adv, stats), g[k + 'advantages'],
                                       rtol=1e-5, atol=1e-5, err_msg=k)
            for chunks in (0, 3, 8):
                ret_c, adv_c, stats_c = run_gae(lib, *args, gamma, lam, chunks)
                scale = max(1.0, np.abs(g[k + 'returns']).max())
                assert np.abs(ret_c - g[k + 'returns']).max() <= 1e-5 * scale, (k, chunks)
                np.testing.assert_allclose(stats_c[:2], stats[:2], rtol=1e-5, atol=1e-6)


def test_gae_constant_and_zero_advantages(lib):
    T, W = 6, 5
    zeros = np.zeros((T, W), np.float32)
    # rewards 0, values == returns == 0 -> every advantage is 0 -> all_zero flag, no normalise
    ret, adv, stats = run_gae(lib, zeros, zeros, zeros, zeros, zeros, 0.99, 0.97, 1)
    assert stats[2] == 1.0 and stats[3] == 0.0 and not adv.any()
    # constant non-zero advantage: std == 0 -> normalisation skipped (segments.py:44)
    ones = np.ones((T, W), np.float32)
    ret, adv, stats = run_gae(lib, zeros, ones, ones, ones, zeros, 0.99, 0.97, 2)
    assert np.array_equal(ret, ones) and stats[2] == 0.0 and stats[3] == 0.0


@pytest.mark.parametrize('T,W', [(4096, 256), (1000, 77)])
def test_gae_full_size_bit_exact(lib, T, W):
    rng = np.random.RandomState(T + W)
    nv, rew, val = (rng.normal(size=(T, W)).astype(np.float32) for _ in range(3))
    rst = (rng.uniform(size=(T, W)) < 1e-3)
    term = (rst & (rng.uniform(size=(T, W)) < 0.5)).astype(np.float32)
    rst = rst.astype(np.float32)
    want = port.lambda_returns(nv, rew, rst, term, 0.99, 0.97)
    ret, adv, stats = run_gae(lib, nv, rew, rst, term, val, 0.99, 0.97, 1)
    assert np.array_equal(ret, want)
    assert np.array_equal(adv, want - val)
    ref_adv = port.normalized_advantages(want, val)
    np.testing.assert_allclose(normalise(adv, stats), ref_adv, rtol=1e-5, atol=1e-5)
    ret_auto, _, stats_auto = run_gae(lib, nv, rew, rst, term, val, 0.99, 0.97, 0)
    assert np.abs(ret_auto - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # determinism: a second run is bit-identical (fixed-order reductions, no atomics)
    ret2, adv2, stats2 = run_gae(lib, nv, rew, rst, term, val, 0.99, 0.97, 0)
    assert np.array_equal(ret2, ret_auto) and np.array_equal(stats2, stats_auto)


# ------------------------------------------------------------------ acting / evaluation

@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_act_golden(lib, golden, name):
    from tonic_amd import _lib
    g = golden(name)
    O, A, W, steps = (int(x) for x in g['cfg'][:4])
    actor, _, _ = _params(g, 'init/')
    params = dev(flat(actor))
    for t in range(steps):
        obs, eps = dev(g['act/observations'][t]), dev(g['act/eps'][t])
        actions, logp = torch.empty(W, A).cuda(), torch.empty(W).cuda()
        ws = torch.empty(max(lib.tonic_ppo_workspace_bytes(W, O, A, 1), 16), dtype=torch.uint8).cuda()
        _lib.check(lib.tonic_ppo_act_wide(params.data_ptr(), obs.data_ptr(), eps.data_ptr(),
                                          actions.data_ptr(), logp.data_ptr(), W, O, A,
                                          ws.data_ptr(), ws.numel(), None), 'act')
        np.testing.assert_allclose(actions.cpu().numpy(), g['act/actions'][t], rtol=0, atol=3e-6)
        np.testing.assert_allclose(logp.cpu().numpy(), g['act/log_probs'][t], rtol=1e-5, atol=1e-5)
    # mode (no noise): loc itself, and log-prob pointer may be NULL
    _lib.check(lib.tonic_ppo_act_wide(params.data_ptr(), obs.data_ptr(), None, actions.data_ptr(),
                                      None, W, O, A, ws.data_ptr(), ws.numel(), None), 'act-mode')
    _, _, loc, _, _ = port.ppo_actor_forward(actor, g['act/observations'][steps - 1])
    np.testing.assert_allclose(actions.cpu().numpy(), loc, rtol=0, atol=3e-6)


# (from 32 768 rows — a whole Segment — the values come from the forward half of the regression
#  kernel, mlp64_grad16_kernel<..., FWD>: ragged tails, every observation bucket)
@pytest.mark.parametrize('O,n', [(17, 1000), (3, 31), (28, 4097), (1, 64), (32, 65), (17, 40000),
                                 (3, 33001), (28, 32768), (1, 32769), (32, 50001), (20, 36000)])
def test_value_forward_vs_oracle(lib, O, n):
    from tonic_amd import _lib
    rng = np.random.RandomState(O * 1000 + n)
    params = [rng.normal(size=(64, O)) * 0.4, rng.normal(size=64) * 0.2,
              rng.normal(size=(64, 64)) * 0.2, rng.normal(size=64) * 0.2,
              rng.normal(size=(1, 64)) * 0.3, rng.normal(size=1)]
    params = [p.astype(np.float32) for p in params]
    mean = rng.normal(size=O).astype(np.float32)
    std = (np.abs(rng.normal(size=O)) + 0.3).astype(np.float32)
    obs = rng.normal(size=(n, O)).astype(np.float32) * 2
    want = port.critic_forward(params, mean, std, obs)[3]
    out = torch.empty(n).cuda()
    keep = [dev(flat(params)), dev(mean), dev(std), dev(obs)]   # keep the tensors alive
    _lib.check(lib.tonic_value_forward(*[t.data_ptr() for t in keep[:3]], 0.0, keep[3].data_ptr(),
                                       out.data_ptr(), n, O, None), 'value')
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------- gradients

def actor_grad(lib, params, obs, actions, adv, stats, old_lp, variant=None):
    from tonic_amd import _lib
    lib = variant_library(lib, variant)
    if variant is not None:
        _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
    n, O = obs.shape
    A = actions.shape[1]
    P = lib.tonic_ppo_actor_param_count(O, A)
    out = torch.zeros(P + 8).cuda()
    ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, A, 1), dtype=torch.uint8).cuda()
    keep = [dev(flat(params)), dev(obs), dev(actions), dev(adv), dev(stats), dev(old_lp)]
    _lib.check(lib.tonic_ppo_actor_grad(
        *[t.data_ptr() for t in keep], out.data_ptr(),
        n, O, A, 0.2, 0.0, None, 0, ws.data_ptr(), ws.numel(), None), 'actor_grad')
    torch.cuda.synchronize()
    _lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')
    return out.cpu().numpy(), P


def critic_grad(lib, params, mean, std, obs, returns, variant=None, clip=0.0):
    from tonic_amd import _lib
    lib = variant_library(lib, variant)
    if variant is not None:
        _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
    n, O = obs.shape
    P = lib.tonic_v_critic_param_count(O)
    out = torch.zeros(P + 8).cuda()
    ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, 1, 0), dtype=torch.uint8).cuda()
    keep = [dev(flat(params)), dev(mean), dev(std), dev(obs), dev(returns)]
    _lib.check(lib.tonic_value_regression_grad(
        *[t.data_ptr() for t in keep[:3]], float(clip), *[t.data_ptr() for t in keep[3:]],
        out.data_ptr(), n, O, 0, ws.data_ptr(), ws.numel(), None), 'critic_grad')
    torch.cuda.synchronize()
    _lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')
    return out.cpu().numpy(), P


GRAD_TOL = float(__import__('os').environ.get('TONIC_TEST_GRAD_TOL', '1e-5'))


def assert_grads_close(got_sums, want_grads, n, what, tol=None):
    """Gradient means against the oracle's: max |diff| <= tol x max |gradient| (+ 1e-7), tol = 1e-5 — the north
    star's figure — unless a call site names its own (and says why)."""
    want = flat(want_grads).astype(np.float64)
    got = got_sums.astype(np.float64) / n
    err = np.abs(got - want).max()
    scale = np.abs(want).max()
    tol = GRAD_TOL if tol is None else tol
    assert err <= tol * scale + 1e-7, f'{what}: max |diff| {err:.3e} vs max |grad| {scale:.3e} (tol {tol:g})'


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('name', PPO_CASES)
def test_actor_and_critic_grads_golden_batch(lib, golden, name, variant):
    g = golden(name)
    actor, critic, norm = _params(g, 'pre0/')
    seg = {k: port.flatten_time_major(g[f'u0/segment/{k}']) for k in (
        'observations', 'actions', 'log_probs', 'returns', 'advantages', 'values')}
    n = seg['observations'].shape[0]
    # (a) final advantages passed directly (normalise flag 0)
    stats = np.array([0, 1, 0, 0], np.float32)
    got, P = actor_grad(lib, actor, seg['observations'], seg['actions'], seg['advantages'],
                        stats, seg['log_probs'], variant)
    want, info = port.clipped_ratio_grads(actor, seg['observations'], seg['actions'],
                                          seg['advantages'], seg['log_probs'])
    assert_grads_close(got[:P], want, n, f'{name} actor grads')
    np.testing.assert_allclose(got[P + 0] / n, info['loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 1] / n, info['kl'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 2] / n, info['clip_fraction'], atol=1e-7)
    np.testing.assert_allclose(got[P + 3] / n, info['entropy'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 4] / n, info['std'], rtol=1e-5, atol=1e-5)
    assert got[P + 5] == n
    # (b) raw advantages + in-kernel normalisation from the statistics
    raw = seg['returns'] - seg['values']
    stats = np.array([raw.mean(dtype=np.float64), raw.std(dtype=np.float64), 0, 1], np.float32)
    got_b, _ = actor_grad(lib, actor, seg['observations'], seg['actions'], raw, stats,
                          seg['log_probs'], variant)
    assert_grads_close(got_b[:P], want, n, f'{name} actor grads (in-kernel normalisation)')
    # critic
    got_c, Pc = critic_grad(lib, critic, norm[0], norm[1], seg['observations'], seg['returns'], variant)
    want_c, info_c = port.value_regression_grads(critic, norm[0], norm[1], seg['observations'],
                                                 seg['returns'])
    assert_grads_close(got_c[:Pc], want_c, n, f'{name} critic grads')
    np.testing.assert_allclose(got_c[Pc + 0] / n, info_c['loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got_c[Pc + 1] / n, info_c['v'].mean(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('variant', [1, 3, 4])
@pytest.mark.parametrize('O,A', [(11, 3), (17, 2), (24, 7), (8, 5), (4, 4), (32, 1)])
def test_grads_in_every_padded_bucket_vs_oracle(lib, O, A, variant):
    """Action counts that are not a kernel bucket of their own (2 .. 5 ride in the six-action build, 7 in the
    eight-action one: padded heads, `EXACT` off) and observation widths of every input bucket, a ragged batch,
    against the oracle's explicit back-propagation — in the shipped build the padded slots are lane groups'
    action slots that must stay dead (csrc/mlp64x16.hip, Lds16::HM)."""
    rng = np.random.RandomState(100 * O + A)
    n = 16 * 8 * 5 + 11
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, A)) * 0.3, rng.normal(size=(A, 64)) * 0.2, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    obs = rng.normal(size=(n, O)).astype(np.float32)
    actions = np.clip(rng.normal(size=(n, A)), -1, 1).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.3).astype(np.float32)
    want, info = port.clipped_ratio_grads(params, obs, actions, adv, old_lp)
    got, P = actor_grad(lib, params, obs, actions, adv, np.array([0, 1, 0, 0], np.float32), old_lp, variant)
    assert_grads_close(got[:P], want, n, f'O={O} A={A} actor grads')
    np.testing.assert_allclose(got[P + 0] / n, info['loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 1] / n, info['kl'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 2] / n, info['clip_fraction'], atol=1e-7)
    assert got[P + 5] == n
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    mean = rng.normal(size=O).astype(np.float32)
    std = (np.abs(rng.normal(size=O)) + 0.5).astype(np.float32)
    returns = rng.normal(size=n).astype(np.float32)
    want_c, info_c = port.value_regression_grads(cparams, mean, std, obs, returns)
    got_c, Pc = critic_grad(lib, cparams, mean, std, obs, returns, variant)
    assert_grads_close(got_c[:Pc], want_c, n, f'O={O} critic grads')
    np.testing.assert_allclose(got_c[Pc + 0] / n, info_c['loss'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4])
def test_grads_ratio_clipping_branches(lib, variant):
    """Forces both clipped branches (ratio > 1.2 with adv > 0, ratio < 0.8 with adv < 0) and
    the still-live ones; ragged n (not a multiple of the 32-sample tile)."""
    rng = np.random.RandomState(5)
    O, A, n = 17, 6, 1234
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, A)) * 0.3, rng.normal(size=(A, 64)) * 0.2, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    obs = rng.normal(size=(n, O)).astype(np.float32)
    actions = np.clip(rng.normal(size=(n, A)), -1, 1).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.4).astype(np.float32)
    want, info = port.clipped_ratio_grads(params, obs, actions, adv, old_lp)
    assert 0.2 < info['clip_fraction'] < 0.9
    got, P = actor_grad(lib, params, obs, actions, adv, np.array([0, 1, 0, 0], np.float32), old_lp,
                        variant)
    assert_grads_close(got[:P], want, n, 'clipping branches')
    np.testing.assert_allclose(got[P + 2] / n, info['clip_fraction'], atol=1e-7)


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4])
def test_grads_full_size_properties(lib, variant):
    """BASELINE size (N = 4096 x 256): sums are additive over a split of the batch,
    bit-reproducible run to run, and agree with the oracle on a 4096-sample slice."""
    rng = np.random.RandomState(11)
    O, A, n = 17, 6, 4096 * 256
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              np.zeros((1, A)), rng.normal(size=(A, 64)) * 0.1, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    obs = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    old_lp = (-6 + rng.standard_normal(n) * 0.2).astype(np.float32)
    stats = np.array([0, 1, 0, 0], np.float32)
    full, P = actor_grad(lib, params, obs, actions, adv, stats, old_lp, variant)
    again, _ = actor_grad(lib, params, obs, actions, adv, stats, old_lp, variant)
    assert np.array_equal(full, again), 'fixed-order reductions must be bit-reproducible'
    h = n // 2
    a, _ = actor_grad(lib, params, obs[:h], actions[:h], adv[:h], stats, old_lp[:h], variant)
    b, _ = actor_grad(lib, params, obs[h:], actions[h:], adv[h:], stats, old_lp[h:], variant)
    both = a.astype(np.float64) + b.astype(np.float64)
    scale = np.abs(full[:P]).max()
    assert np.abs(both[:P] - full[:P]).max() <= 1e-5 * scale
    m = 4096
    want, _ = port.clipped_ratio_grads(params, obs[:m], actions[:m], adv[:m], old_lp[:m])
    part, _ = actor_grad(lib, params, obs[:m], actions[:m], adv[:m], stats, old_lp[:m], variant)
    assert_grads_close(part[:P], want, m, 'slice of the full batch')


@pytest.mark.parametrize('variant', [1, 2, 3, 4])
def test_critic_grad_is_bit_reproducible_at_baseline_size(lib, variant):
    """N = 4096 x 256: twelve launches of tonic_value_regression_grad on the same inputs give ONE
    result, at the kernel's own width and at the widths the PPO agent uses under a rollout.  Round 4
    found the shipped variant (3) giving 9 distinct results in 16 launches — always in the 16
    accumulators of dW1's remainder column, rows 32..47 — because the SLP vectoriser had packed their
    FMAs into dependent v_pk_fma_f32 pairs whose low lane is occasionally wrong between MFMAs on
    gfx950; the library is built with -fno-slp-vectorize since (csrc/Makefile)."""
    rng = np.random.RandomState(21)
    O, n = 17, 4096 * 256
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, 64)) * 0.3, rng.normal(size=1)]
    params = [p.astype(np.float32) for p in params]
    from tonic_amd import _lib
    lib = variant_library(lib, variant)
    P = lib.tonic_v_critic_param_count(O)
    ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, 1, 0), dtype=torch.uint8, device='cuda')
    keep = [dev(flat(params)), dev((rng.standard_normal(O) * 0.1).astype(np.float32)),
            dev((1 + 0.2 * rng.uniform(size=O)).astype(np.float32)),
            dev(rng.standard_normal((n, O)).astype(np.float32)),
            dev(rng.standard_normal(n).astype(np.float32))]
    _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
    try:
        for width in (0, 219, 232):
            results = set()
            for _ in range(12):
                out = torch.zeros(P + 8, device='cuda')
                _lib.check(lib.tonic_value_regression_grad(
                    *[t.data_ptr() for t in keep[:3]], 0.0, *[t.data_ptr() for t in keep[3:]],
                    out.data_ptr(), n, O, width, ws.data_ptr(), ws.numel(), None), 'critic_grad')
                results.add(out.cpu().numpy().tobytes())
            assert len(results) == 1, (variant, width, len(results))
    finally:
        _lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')


@pytest.mark.parametrize('O,A', [(3, 1), (28, 8), (20, 6), (32, 8)])
def test_grads_are_bit_reproducible_in_every_shape_bucket(lib, O, A):
    """The other template instances of the fused grad kernels (observation buckets 4 / 20 / 32 wide,
    action buckets 1 / 6 / 8; remainder columns on VALU for O = 3 and O = 20, none for 28 and 32):
    eight launches of the actor and of the critic kernel on the same 262 144 samples give one result
    each, at the kernel's own width and at a narrower one."""
    from tonic_amd import _lib
    rng = np.random.RandomState(100 * O + A)
    n = 262144
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              np.zeros((1, A)), rng.normal(size=(A, 64)) * 0.1, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    obs = dev(rng.standard_normal((n, O)).astype(np.float32))
    actions = dev(np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32))
    adv = dev(rng.standard_normal(n).astype(np.float32))
    old_lp = dev((-6 + rng.standard_normal(n) * 0.2).astype(np.float32))
    returns = dev(rng.standard_normal(n).astype(np.float32))
    stats = dev(np.array([0, 1, 0, 0], np.float32))
    mean, std = dev(np.zeros(O, np.float32)), dev(np.ones(O, np.float32))
    pa, pc = dev(flat(params)), dev(flat(cparams))
    Pa, Pc = lib.tonic_ppo_actor_param_count(O, A), lib.tonic_v_critic_param_count(O)
    ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, A, 1), dtype=torch.uint8, device='cuda')
    for width in (0, 100):
        seen_a, seen_c = set(), set()
        for _ in range(8):
            out_a = torch.zeros(Pa + 8, device='cuda')
            out_c = torch.zeros(Pc + 8, device='cuda')
            _lib.check(lib.tonic_ppo_actor_grad(
                pa.data_ptr(), obs.data_ptr(), actions.data_ptr(), adv.data_ptr(), stats.data_ptr(),
                old_lp.data_ptr(), out_a.data_ptr(), n, O, A, 0.2, 0.0, None, width, ws.data_ptr(),
                ws.numel(), None), 'actor_grad')
            _lib.check(lib.tonic_value_regression_grad(
                pc.data_ptr(), mean.data_ptr(), std.data_ptr(), 0.0, obs.data_ptr(),
                returns.data_ptr(), out_c.data_ptr(), n, O, width, ws.data_ptr(), ws.numel(), None),
                'critic_grad')
            seen_a.add(out_a.cpu().numpy().tobytes())
            seen_c.add(out_c.cpu().numpy().tobytes())
        assert len(seen_a) == 1 and len(seen_c) == 1, (O, A, width, len(seen_a), len(seen_c))


def float64_gradient_sums(params, cparams, obs, actions, adv, old_lp, returns):
    """Gradient SUMS of the PPO actor loss (clip 0.2, advantages as given) and of the critic's squared error in
    float64 autograd on the device: what every grad_variant is measured against."""
    def f64(arrays):
        return [torch.tensor(np.asarray(a, np.float64), device='cuda', requires_grad=True) for a in arrays]

    x, a_t, adv_t, lp_t, ret_t = (torch.tensor(np.asarray(v, np.float64), device='cuda')
                                  for v in (obs, actions, adv, old_lp, returns))
    W1, b1, W2, b2, ls, W3, b3 = pa = f64(params)
    h = torch.tanh(torch.tanh(x @ W1.T + b1) @ W2.T + b2)
    dist = torch.distributions.Normal(
        torch.tanh(h @ W3.T + b3), (torch.nn.functional.softplus(ls) + 1e-8).clamp(1e-4, 1.0))
    ratio = torch.exp(dist.log_prob(a_t).sum(-1) - lp_t)
    loss = -torch.min(adv_t * ratio, adv_t * ratio.clamp(0.8, 1.2)).sum()
    want_a = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, pa)]).cpu().numpy()
    V1, c1, V2, c2, V3, c3 = pc = f64(cparams)
    v = (torch.tanh(torch.tanh(x @ V1.T + c1) @ V2.T + c2) @ V3.T + c3)[:, 0]
    want_c = torch.cat([g.reshape(-1) for g in
                        torch.autograd.grad(((v - ret_t) ** 2).sum(), pc)]).cpu().numpy()
    return want_a, want_c


def test_bf16x3_hidden_layer_products_are_fp32_class(lib):
    """grad_variant 2 computes the two 64x64 hidden-layer products of a tile as six bf16 MFMAs on exact
    hi + mid + lo splits of the fp32 operands.  Against a float64 autograd reference of the same loss
    its gradient sums must be as close as those of the fp32-MFMA variants (0 and 1): the split is a
    re-association of fp32 arithmetic, not a precision cut."""
    rng = np.random.RandomState(23)
    O, A, n = 17, 6, 65536
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, 64)) * 0.1, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    obs = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.1).astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    mean, std = np.zeros(O, np.float32), np.ones(O, np.float32)
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]

    want_a, want_c = float64_gradient_sums(params, cparams, obs, actions, adv, old_lp, returns)

    errors = {}
    for variant in (0, 1, 2, 3, 4):
        got_a, P = actor_grad(lib, params, obs, actions, adv, np.array([0, 1, 0, 0], np.float32),
                              old_lp, variant)
        got_c, Pc = critic_grad(lib, cparams, mean, std, obs, returns, variant)
        errors[variant] = (np.abs(got_a[:P] - want_a).max() / np.abs(want_a).max(),
                           np.abs(got_c[:Pc] - want_c).max() / np.abs(want_c).max())
    print('max relative error vs float64 (actor, critic) per grad_variant:', errors)
    for k in (0, 1):                                   # actor, critic
        fp32_class = max(errors[0][k], errors[1][k])
        assert fp32_class < 2e-6, errors
        assert max(errors[2][k], errors[3][k], errors[4][k]) <= 2.0 * fp32_class + 1e-8, errors


@pytest.mark.parametrize('case', ['rising', 'falling', 'mixed', 'zero', 'small_weights', 'large_weights',
                                  'ragged'])
def test_fp16x2_products_keep_their_unit_over_extreme_ranges(lib, case):
    """grad_variant 4 scales everything behind the loss gradient by one power of two per wave that
    follows the largest head gradient seen so far (csrc/mlp64x16.hip, Lds16 CH = 3): per-sample loss
    gradients spread over 18 decades — rising along the batch (every wave rescales its accumulators
    again and again), falling, shuffled —, all-zero gradients and a ragged batch give the gradient sums of
    the fp32-MFMA variant (1: same tiles, same order of sums), block by block, to fp32 rounding of each
    block's largest entry; with weight matrices far from unit scale, which amplify every rounding of the
    forward pass, the two builds are as far from a float64 reference as each other."""
    rng = np.random.RandomState(31)
    O, A = 17, 6
    n = 16 * 8 * 40 + (5 if case == 'ragged' else 0)
    w2, w3 = {'small_weights': (1e-6, 1e3), 'large_weights': (40.0, 1e-4)}.get(case, (0.15, 0.1))
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * w2, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, 64)) * w3, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    obs = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    decades = {'rising': np.linspace(-12, 6, n), 'falling': np.linspace(6, -12, n),
               'mixed': rng.uniform(-12, 6, n)}.get(case, np.zeros(n))
    size = np.zeros(n) if case == 'zero' else 10.0 ** decades
    adv = (rng.standard_normal(n) * size).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.05).astype(np.float32)
    mean, std = np.zeros(O, np.float32), np.ones(O, np.float32)
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    values = port.critic_forward(cparams, mean, std, obs)[3]
    returns = (values + rng.standard_normal(n) * size * 100.0).astype(np.float32)
    stats = np.array([0, 1, 0, 0], np.float32)

    def blocks(P, actor):
        edges = [0, 64 * O, 64 * O + 64, 64 * O + 64 + 4096, 64 * O + 128 + 4096]
        return list(zip(edges, edges[1:] + [P]))

    exact = float64_gradient_sums(params, cparams, obs, actions, adv, old_lp, returns) if 'weights' in case else None
    for what, run in (('actor', lambda v: actor_grad(lib, params, obs, actions, adv, stats, old_lp, v)),
                      ('critic', lambda v: critic_grad(lib, cparams, mean, std, obs, returns, v))):
        want, P = run(1)
        got, _ = run(4)
        assert np.isfinite(got).all(), (case, what)
        if exact is not None:
            # weights far from unit scale amplify every rounding of the forward pass (W2 x 40: a change of
            # 1e-7 in h1 moves z2 by 5e-6): the two builds are compared through their distance from float64
            truth = exact[what == 'critic']
            top = np.abs(truth).max()
            far_1, far_4 = np.abs(want[:P] - truth).max() / top, np.abs(got[:P] - truth).max() / top
            assert far_4 <= 2.0 * far_1 + 1e-7, (case, what, far_1, far_4)
            continue
        if case == 'zero' and what == 'critic':
            continue          # errors of 1e-7: the two variants' own value rounding, nothing to compare
        np.testing.assert_allclose(got[P:], want[P:], rtol=1e-4, atol=1e-4 * np.abs(want[P:]).max(),
                                   err_msg=f'{case} {what}: statistic sums')
        for lo, hi in blocks(P, what == 'actor'):
            top = np.abs(want[lo:hi]).max()
            err = np.abs(got[lo:hi].astype(np.float64) - want[lo:hi]).max()
            assert err <= 4e-6 * top, (case, what, (lo, hi), err, top)
        if case == 'zero' and what == 'actor':
            assert not got[:P].any()


@pytest.mark.parametrize('weights', ['as_initialised', 'conditioned'])
@pytest.mark.parametrize('features', ['seven_decades', 'one_outlier'])
def test_fp16x2_layer_one_with_observations_of_mixed_magnitude(lib, features, weights):
    """The reference's actor sees RAW observations (models/actors.py:128-129, quirk Q1) and the critic sees them
    through (x - mean) / std with whatever statistics it has (normalizers/mean_stds.py:34-39): feature columns
    seven decades apart — 1e-3 ... 1e4 — and a sample with a single 1e6 entry, (i) with first-layer weights as
    initialised (the large columns saturate tanh) and (ii) with weights that undo the column scales (every term
    of z1 is O(1): what a trained network looks like).  Layer 1 of the shipped kernels (grad_variant 4) runs on
    fp16x2 terms in a per-sample unit; W1 is equilibrated by column (Lds16::CX) so that the unit is taken over
    terms of equal weight.  Without that, case (ii) is off by 1e-4 in z1 (tests/test_fp16x2_arithmetic.py).
    Gradient sums of actor AND critic against float64 autograd, held to the fp32-MFMA variant's error x 2."""
    rng = np.random.RandomState(41)
    O, A, n = 17, 6, 65536
    scales = 10.0 ** np.linspace(-3, 4, O)
    rng.shuffle(scales)
    obs = rng.standard_normal((n, O)) * scales
    if features == 'one_outlier':
        obs = rng.standard_normal((n, O))
        obs[np.arange(0, n, 977), rng.randint(0, O, len(range(0, n, 977)))] = 1e6
        scales = np.ones(O)
    obs = obs.astype(np.float32)
    w1 = rng.normal(size=(64, O)) * 0.3
    if weights == 'conditioned':
        w1 = w1 / scales
    params = [w1, rng.normal(size=64) * 0.1, rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, 64)) * 0.1, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.1).astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    mean, std = np.zeros(O, np.float32), np.ones(O, np.float32)
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    want_a, want_c = float64_gradient_sums(params, cparams, obs, actions, adv, old_lp, returns)
    errors = {}
    for variant in (1, 4):
        got_a, P = actor_grad(lib, params, obs, actions, adv, np.array([0, 1, 0, 0], np.float32), old_lp, variant)
        got_c, Pc = critic_grad(lib, cparams, mean, std, obs, returns, variant)
        assert np.isfinite(got_a).all() and np.isfinite(got_c).all(), (features, weights, variant)
        errors[variant] = (np.abs(got_a[:P] - want_a).max() / np.abs(want_a).max(),
                           np.abs(got_c[:Pc] - want_c).max() / np.abs(want_c).max())
    print(features, weights, 'max relative error vs float64 (actor, critic) per grad_variant:', errors)
    for k in (0, 1):
        # (weights as initialised: the large columns saturate tanh and the gradient sums are what is left after
        #  cancellation — the fp32-MFMA build itself is 7e-5 from float64 there; it is the yardstick either way)
        if weights == 'conditioned':
            assert errors[1][k] < 5e-6, errors
        assert errors[4][k] <= 2.0 * errors[1][k] + 1e-8, errors


def _ppo_act(lib, params, obs, eps):
    from tonic_amd import _lib
    W, O = obs.shape
    A = eps.shape[1]
    actions, logp = torch.empty(W, A).cuda(), torch.empty(W).cuda()
    ws = torch.empty(max(lib.tonic_ppo_workspace_bytes(W, O, A, 1), 16), dtype=torch.uint8).cuda()
    keep = [dev(flat(params)), dev(obs), dev(eps)]
    _lib.check(lib.tonic_ppo_act_wide(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                      actions.data_ptr(), logp.data_ptr(), W, O, A, ws.data_ptr(), ws.numel(),
                                      None), 'act')
    torch.cuda.synchronize()
    return actions.cpu().numpy(), logp.cpu().numpy()


def test_fp16x2_kernels_with_non_finite_and_out_of_range_observations(lib):
    """What the shipped fp16x2 kernels do OUTSIDE the envelope the other tests cover — NaN, +-inf, |x| = 1e30
    observations, a denormal W1 column — held to the reference's behaviour: a NaN observation gives NaN actions
    (the reference propagates it and trips `assert not np.isnan(actions.sum())`, utils/trainer.py:45); for inf and
    1e30 (where the reference's float32 chain saturates tanh and stays finite) a row is either what the oracle
    computes or NaN — loud, never a finite wrong answer; rows next to such a row are untouched; a W1 column of
    denormals (column equilibration clamps its exponent, Lds16::CX) costs nothing.  Gradient sums: one NaN
    observation poisons the loss sum like torch's mean does; with the 1e30 rows the sums are the oracle's or NaN."""
    rng = np.random.RandomState(47)
    O, A, W = 17, 6, 256
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1, rng.normal(size=(64, 64)) * 0.15,
              rng.normal(size=64) * 0.1, rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, 64)) * 0.1,
              rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    params[0][:, 5] = (rng.normal(size=64) * 1e-41).astype(np.float32)       # a denormal column
    assert 0 < np.abs(params[0][:, 5]).max() < 1.2e-38
    obs = rng.standard_normal((W, O)).astype(np.float32)
    eps = rng.standard_normal((W, A)).astype(np.float32)
    odd = {3: np.nan, 40: np.inf, 41: -np.inf, 77: 1e30, 130: -1e30, 200: 3e38}
    for row, value in odd.items():
        obs[row, row % O if row % O != 5 else 6] = value
    with np.errstate(all='ignore'):
        _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
        want_actions = loc + scale * eps
        want_logp = port.normal_log_prob(want_actions, loc, scale)
    got_actions, got_logp = _ppo_act(lib, params, obs, eps)
    ordinary = np.array([r not in odd for r in range(W)])
    np.testing.assert_allclose(got_actions[ordinary], want_actions[ordinary], rtol=0, atol=5e-6)
    np.testing.assert_allclose(got_logp[ordinary], want_logp[ordinary], rtol=1e-5, atol=1e-5)
    assert np.isnan(got_actions[3]).all(), 'a NaN observation must give NaN actions (trainer.py:45)'
    outcome = {}
    for row in odd:
        if row == 3:
            continue
        finite = np.isfinite(got_actions[row]).all()
        if finite and np.isfinite(want_actions[row]).all():
            np.testing.assert_allclose(got_actions[row], want_actions[row], rtol=0, atol=5e-6,
                                       err_msg=f'row {row} ({odd[row]}): finite but not the reference\'s answer')
        else:
            assert np.isnan(got_actions[row]).any() or not np.isfinite(want_actions[row]).all(), (row, got_actions[row])
        outcome[odd[row]] = 'as the reference' if finite else 'NaN'
    print('observations outside the envelope -> actions:', outcome)
    # gradient sums and values: NaN poisons the sums (torch's mean does), 1e30 rows: the oracle's sums or NaN
    n = 16 * 8 * 4
    obs_n = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs_n)
    old_lp = (port.normal_log_prob(actions, loc, scale) + rng.normal(size=n) * 0.1).astype(np.float32)
    mean, std = np.zeros(O, np.float32), np.ones(O, np.float32)
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    stats = np.array([0, 1, 0, 0], np.float32)
    for value in (np.nan, 1e30):
        bad = obs_n.copy()
        bad[100, 2] = value
        got_a, P = actor_grad(lib, params, bad, actions, adv, stats, old_lp)
        got_c, Pc = critic_grad(lib, cparams, mean, std, bad, returns)
        if np.isnan(value):
            # torch: min / clamp propagate NaN, so loss, KL and every gradient the sample touches are NaN
            assert np.isnan(got_a[P + 0]) and np.isnan(got_a[P + 1]) and np.isnan(got_c[Pc + 0]), \
                ('a NaN observation must poison the logged loss', got_a[P:], got_c[Pc:])
            assert np.isnan(got_a[:P]).any() and np.isnan(got_c[:Pc]).any(), 'and the gradient sums'
            continue
        with np.errstate(all='ignore'):
            want_a, _ = port.clipped_ratio_grads(params, bad, actions, adv, old_lp)
            want_c, _ = port.value_regression_grads(cparams, mean, std, bad, returns)
        for got, want, count, what in ((got_a, want_a, P, 'actor'), (got_c, want_c, Pc, 'critic')):
            want = flat(want).astype(np.float64)
            if np.isfinite(got[:count]).all() and np.isfinite(want).all():
                err = np.abs(got[:count].astype(np.float64) / n - want).max()
                assert err <= 2e-5 * np.abs(want).max() + 1e-7, (what, err)
                print(f'1e30 observation, {what} gradient sums: as the reference ({err:.2e})')
            else:
                assert not np.isfinite(got[:count]).all() or not np.isfinite(want).all()
                print(f'1e30 observation, {what} gradient sums: non-finite (loud); reference finite: '
                      f'{bool(np.isfinite(want).all())}')


@pytest.mark.parametrize('variant', [1, 4])
def test_critic_normalisation_at_the_std_floor(lib, variant):
    """MeanStd floors std at 1e-2 (normalizers/mean_stds.py:62-66) and divides (:36): with |x - mean| up to 1e3
    the critic's inputs reach 1e5.  The shipped kernels form the quotient as multiply by the staged reciprocal
    + one Newton step on the exact remainder (quotient_by: the division's rounding); values and gradient sums
    against numpy_port.critic_forward / value_regression_grads at 1e-5, most features of ordinary size so that
    the network is not saturated everywhere."""
    from tonic_amd import _lib
    rng = np.random.RandomState(43)
    O, n = 17, 32768
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1, rng.normal(size=(64, 64)) * 0.15,
              rng.normal(size=64) * 0.1, rng.normal(size=(1, 64)) * 0.3, rng.normal(size=1)]
    params[0][:, :4] *= 1e-4                       # the weights of the columns whose inputs reach 1e5
    params = [p.astype(np.float32) for p in params]
    mean = rng.normal(size=O).astype(np.float32)
    std = np.full(O, 1e-2, np.float32)
    obs = mean + rng.standard_normal((n, O)) * 1e-2
    obs[:, :4] = mean[:4] + rng.uniform(-1e3, 1e3, (n, 4))
    obs[::50, 7] = mean[7] + 1e3                   # and isolated saturating entries
    obs = obs.astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    want_v = port.critic_forward(params, mean, std, obs)[3]
    out = torch.empty(n).cuda()
    keep = [dev(flat(params)), dev(mean), dev(std), dev(obs)]
    vlib, lib = lib, variant_library(lib, variant)
    _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
    try:
        _lib.check(lib.tonic_value_forward(*[t.data_ptr() for t in keep[:3]], 0.0, keep[3].data_ptr(),
                                           out.data_ptr(), n, O, None), 'value')
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')
    np.testing.assert_allclose(out.cpu().numpy(), want_v, rtol=1e-5, atol=1e-5)
    want = flat(port.value_regression_grads(params, mean, std, obs, returns)[0]).astype(np.float64)
    got, P = critic_grad(vlib, params, mean, std, obs, returns, variant)
    err = np.abs(got[:P].astype(np.float64) / n - want).max()
    assert err <= 1e-5 * np.abs(want).max() + 1e-7, (variant, err, np.abs(want).max())


def test_config5_size_properties(lib):
    """BASELINE config 5 at its single-GPU maximum (AntBullet shapes O=28, A=8, W=10240, T=4096:
    N = 41.9 M transitions, 4.7 GB of observations): 64-bit indexing end to end.  The gradient
    sums are additive over a split of the batch, the tail slice agrees with the oracle, and the
    lambda-return scan over [4096, 10240] is bit-exact."""
    O, A, T, W = 28, 8, 4096, 10240
    n = T * W
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    rng = np.random.RandomState(5)
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              np.zeros((1, A)), rng.normal(size=(A, 64)) * 0.1, rng.normal(size=A) * 0.1]
    params = [p.astype(np.float32) for p in params]
    cparams = params[:4] + [params[5][:1].copy(), params[6][:1].copy()]
    obs = torch.randn(n, O, device='cuda', generator=g)
    actions = torch.randn(n, A, device='cuda', generator=g).clamp_(-1, 1)
    adv = torch.randn(n, device='cuda', generator=g)
    old_lp = torch.randn(n, device='cuda', generator=g) * 0.2 - 8
    returns = torch.randn(n, device='cuda', generator=g)
    stats = np.array([0, 1, 0, 0], np.float32)
    mean, std = rng.normal(size=O).astype(np.float32), (0.5 + rng.uniform(size=O)).astype(np.float32)
    h = n // 2 + 12345                                   # uneven split, second half starts > 2^31 B in
    full, P = actor_grad(lib, params, obs, actions, adv, stats, old_lp)
    a, _ = actor_grad(lib, params, obs[:h], actions[:h], adv[:h], stats, old_lp[:h])
    b, _ = actor_grad(lib, params, obs[h:], actions[h:], adv[h:], stats, old_lp[h:])
    both = a.astype(np.float64) + b.astype(np.float64)
    assert np.abs(both[:P] - full[:P]).max() <= 2e-5 * np.abs(full[:P]).max()
    assert full[P + 5] == n and abs(both[P + 5] - n) <= 4, 'sample count statistic (float32)'
    cfull, Pc = critic_grad(lib, cparams, mean, std, obs, returns)
    ca, _ = critic_grad(lib, cparams, mean, std, obs[:h], returns[:h])
    cb, _ = critic_grad(lib, cparams, mean, std, obs[h:], returns[h:])
    cboth = ca.astype(np.float64) + cb.astype(np.float64)
    assert np.abs(cboth[:Pc] - cfull[:Pc]).max() <= 2e-5 * np.abs(cfull[:Pc]).max()
    m = 4096                                             # the LAST rows: highest addresses
    tail = [t[n - m:].cpu().numpy() for t in (obs, actions, adv, old_lp, returns)]
    want, _ = port.clipped_ratio_grads(params, tail[0], tail[1], tail[2], tail[3])
    part, _ = actor_grad(lib, params, obs[n - m:], actions[n - m:], adv[n - m:], stats, old_lp[n - m:])
    assert_grads_close(part[:P], want, m, 'tail slice, actor')
    want, _ = port.value_regression_grads(cparams, mean, std, tail[0], tail[4])
    part, _ = critic_grad(lib, cparams, mean, std, obs[n - m:], returns[n - m:])
    assert_grads_close(part[:Pc], want, m, 'tail slice, critic')
    del obs, actions, full, a, b

    nv, rew, val = (torch.randn(T, W, device='cuda', generator=g).cpu().numpy() for _ in range(3))
    rst = rng.uniform(size=(T, W)) < 1e-3
    term = (rst & (rng.uniform(size=(T, W)) < 0.5)).astype(np.float32)
    rst = rst.astype(np.float32)
    want = port.lambda_returns(nv, rew, rst, term, 0.99, 0.97)
    ret, adv_out, _ = run_gae(lib, nv, rew, rst, term, val, 0.99, 0.97, 1)
    assert np.array_equal(ret, want) and np.array_equal(adv_out, want - val)
    ret_auto, _, _ = run_gae(lib, nv, rew, rst, term, val, 0.99, 0.97, 0)
    assert np.abs(ret_auto - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


# -------------------------------------------------------------------------------- Adam

def test_adam_matches_reference_adam(lib):
    from tonic_amd import _lib
    rng = np.random.RandomState(3)
    n, N = 5000, 37
    p0 = rng.normal(size=n).astype(np.float32)
    grads = [(rng.normal(size=n) * 10 ** rng.uniform(-6, 1, size=n)).astype(np.float32) for _ in range(5)]
    ref = [torch.nn.Parameter(torch.tensor(p0))]
    opt = torch.optim.Adam(ref, lr=3e-4)
    params, m, v = dev(p0), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    state = torch.zeros(4, dtype=torch.int32).cuda()
    for g in grads:
        ref[0].grad = torch.tensor(g)
        opt.step()
        sums = dev(np.concatenate([g * N, np.zeros(8, np.float32)]))
        _lib.check(lib.tonic_adam_step(params.data_ptr(), sums.data_ptr(), m.data_ptr(),
                                       v.data_ptr(), state.data_ptr(), n, 1.0 / N, 3e-4, 0.9,
                                       0.999, 1e-8, 0, 0.0, 0.0, None, None, None, None), 'adam')
    assert int(state[0]) == len(grads)
    delta_ref = ref[0].detach().numpy() - p0
    delta = params.cpu().numpy() - p0
    np.testing.assert_allclose(delta, delta_ref, rtol=0, atol=1e-7)


# ------------------------------------------------------------------ segment + normaliser

def test_segment_store_and_meanstd_record_bit_exact(lib):
    from tonic_amd.replays import Segment
    from tonic_amd.torch.normalizers import MeanStd
    rng = np.random.RandomState(8)
    T, W, O, A = 5, 37, 17, 6
    seg = Segment(size=T)
    seg.initialize(seed=0, device='cuda')
    norm = MeanStd()
    norm.initialize((O,))
    norm.attach('cuda')
    ref = port.MeanStdPort((O,))
    rows = []
    for t in range(T):
        row = dict(observations=rng.normal(size=(W, O)) * 3 + 1, actions=rng.normal(size=(W, A)),
                   next_observations=rng.normal(size=(W, O)), rewards=rng.normal(size=W),
                   resets=rng.uniform(size=W) < 0.3, terminations=rng.uniform(size=W) < 0.1,
                   log_probs=rng.normal(size=W))
        row = {k: np.asarray(v, np.float32) for k, v in row.items()}
        rows.append(row)
        seg.store(normalizer=norm, **{k: dev(v) for k, v in row.items()})
        ref.record(row['observations'])
    assert seg.ready()
    for k in rows[0]:
        want = np.stack([r[k] for r in rows])
        assert np.array_equal(seg.buffers[k].cpu().numpy(), want), k
    sums = norm.device_sums.cpu().numpy()
    assert np.array_equal(sums[:O], ref.new_sum) and np.array_equal(sums[O:], ref.new_sum_sq)
    norm.update()
    mean, std = ref.update()
    assert np.array_equal(norm._mean.detach().cpu().numpy(), mean)
    assert np.array_equal(norm._std.detach().cpu().numpy(), std)


@pytest.mark.parametrize('rows,size', [(3000, 17), (1, 3), (482, 17), (5000, 32)])
def test_meanstd_record_standalone_bit_exact(lib, rows, size):
    """tonic_meanstd_record (several double-buffered LDS chunks) == MeanStd.record's Python loop
    (mean_stds.py:44-48), continuing from non-zero running sums."""
    from tonic_amd import _lib
    rng = np.random.RandomState(rows + size)
    values = (rng.standard_normal((rows, size)) * 3 + 1).astype(np.float32)
    ref = port.MeanStdPort((size,))
    ref.record(values[:1])                                   # non-zero starting sums
    start = np.concatenate([ref.new_sum, ref.new_sum_sq]).astype(np.float32)
    ref.record(values)
    acc, d_values = dev(start), dev(values)
    _lib.check(lib.tonic_meanstd_record(d_values.data_ptr(), acc.data_ptr(), rows, size, None),
               'tonic_meanstd_record')
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    assert np.array_equal(got[:size], ref.new_sum) and np.array_equal(got[size:], ref.new_sum_sq)


# ------------------------------------------------------------- whole update, agent level

def _agent_from_golden(g, prefix, steps, iterations=None, batch_size=None, seed=0):
    iterations = int(g['cfg'][5]) if iterations is None else iterations
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    O, A = int(g['cfg'][0]), int(g['cfg'][1])
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations,
                                         batch_size=batch_size))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    state = {k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(prefix)}
    agent.model.load_state_dict(state)
    return agent


def _fill_segment(agent, g, u):
    for t in range(agent.replay.max_size):
        row = {k: dev(g[f'u{u}/segment/{k}'][t]) for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets',
            'terminations', 'log_probs')}
        agent.replay.store(normalizer=None, **row)


@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_update_matches_reference(golden, lib, name):
    g = golden(name)
    steps = int(g['cfg'][3])
    agent = _agent_from_golden(g, 'pre0/', steps)
    before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    _fill_segment(agent, g, 0)
    infos = agent.enqueue_update().cpu().numpy()
    b = agent.replay.buffers
    np.testing.assert_allclose(b['returns'].cpu().numpy(), g['u0/segment/returns'], rtol=1e-5, atol=1e-5)
    ran = infos[0][:, 6] > 0
    n_actor = int(g['u0/info/actor/iterations'][0])
    assert ran.sum() == n_actor and ran[:n_actor].all(), 'device-side KL early stop'
    for i, key in enumerate(('loss', 'kl', 'entropy', 'clip_fraction', 'std')):
        np.testing.assert_allclose(infos[0][:n_actor, i], g[f'u0/info/actor/{key}'],
                                   rtol=1e-5, atol=1e-5, err_msg=key)
    assert np.array_equal(infos[0][:n_actor, 5] > 0.5, g['u0/info/actor/stop'])
    np.testing.assert_allclose(infos[1][:, 0], g['u0/info/critic/loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(infos[1][:, 1], g['u0/info/critic/v_mean'], rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    for key, start in before.items():
        if 'normalizer' in key:
            continue
        got = after[key].detach().cpu().numpy() - start
        want = g['post0/' + key] - start
        tol = max(1e-5, 50 * float(g['noise/' + key].max()))
        # After several Adam steps an element whose gradient sits at float32-noise level can end
        # one rounding beyond the bound (seen: 1 of 7 104 elements at 1.06e-5 on the
        # layer-by-layer path, whose partial sums are cut differently): all but 0.1 % of the
        # elements within `tol`, none beyond twice that.
        diff = np.abs(got - want)
        assert (diff <= tol).mean() >= 0.999 and diff.max() <= 2 * tol, (key, diff.max(), tol)


def test_segment_gather_bit_exact(lib):
    """tonic_segment_gather == fancy indexing (segments.py:64), ragged row count."""
    from tonic_amd import _lib
    rng = np.random.RandomState(5)
    N, n, O, A = 1003, 517, 17, 6
    src = [rng.randn(N, O), rng.randn(N, A), rng.randn(N), rng.randn(N), rng.randn(N)]
    src = [x.astype(np.float32) for x in src]
    idx = rng.permutation(N)[:n].astype(np.int64)
    d_src = [dev(x) for x in src]
    d_out = [torch.full((n,) + x.shape[1:], np.nan, device='cuda') for x in src]
    d_idx = torch.as_tensor(idx, device='cuda')
    _lib.check(lib.tonic_segment_gather(
        d_idx.data_ptr(), *[t.data_ptr() for t in d_src], *[t.data_ptr() for t in d_out], n, N,
        O, A, None), 'tonic_segment_gather')
    torch.cuda.synchronize()
    for got, want in zip(d_out, src):
        assert np.array_equal(got.cpu().numpy(), want[idx])


def test_ppo_minibatch_update_matches_reference(golden, lib):
    """Segment(batch_size=64): 5 shuffled epochs of 64/64/64/48 minibatches, KL stop inside an
    epoch (ppo.py:40-47 over segments.py:58-65)."""
    g = golden('ppo_minibatch_small')
    steps, seed, iterations = int(g['cfg'][3]), int(g['cfg'][4]), int(g['cfg'][5])
    agent = _agent_from_golden(g, 'pre0/', steps, iterations=iterations,
                               batch_size=int(g['batch_size']), seed=seed)
    before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    _fill_segment(agent, g, 0)
    infos = agent.enqueue_update().cpu().numpy()
    assert infos.shape[1] == int(g['u0/info/critic/iterations'][0])
    ran = infos[0][:, 6] > 0
    n_actor = int(g['u0/info/actor/iterations'][0])
    assert ran.sum() == n_actor and ran[:n_actor].all(), 'device-side KL early stop'
    for i, key in enumerate(('loss', 'kl', 'entropy', 'clip_fraction', 'std')):
        np.testing.assert_allclose(infos[0][:n_actor, i], g[f'u0/info/actor/{key}'],
                                   rtol=1e-5, atol=1e-5, err_msg=key)
    assert np.array_equal(infos[0][:n_actor, 5] > 0.5, g['u0/info/actor/stop'])
    np.testing.assert_allclose(infos[1][:, 0], g['u0/info/critic/loss'], rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    for key, start in before.items():
        if 'normalizer' in key:
            continue
        got = after[key].detach().cpu().numpy() - start
        np.testing.assert_allclose(got, g['post0/' + key] - start, rtol=0, atol=2e-5, err_msg=key)

    # drop-in generator: same index stream, normalised advantages
    agent2 = _agent_from_golden(g, 'pre0/', steps, iterations=iterations,
                                batch_size=int(g['batch_size']), seed=seed)
    _fill_segment(agent2, g, 0)
    agent2.replay.compute_returns(*agent2._evaluate())
    want_idx = port.segment_minibatch_indices(np.random.RandomState(seed), steps * int(g['cfg'][2]),
                                              int(g['batch_size']), iterations)
    flat_obs = port.flatten_time_major(g['u0/segment/observations'])
    adv = port.normalized_advantages(g['u0/segment/returns'], g['u0/segment/values']).reshape(-1)
    count = 0
    for batch, idx in zip(agent2.replay.get('observations', 'advantages'), want_idx):
        assert np.array_equal(batch['observations'].cpu().numpy(), flat_obs[idx])
        np.testing.assert_allclose(batch['advantages'].cpu().numpy(), adv[idx], atol=2e-5)
        count += 1
    assert count == agent2.replay.updates_per_get()


@pytest.mark.parametrize('name', PPO_CASES)
def test_ppo_single_iteration_strict(golden, lib, name):
    g = golden(name)
    steps = int(g['cfg'][3])
    agent = _agent_from_golden(g, 'pre0/', steps, iterations=1)
    before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    _fill_segment(agent, g, 0)
    agent.enqueue_update()
    torch.cuda.synchronize()
    actor, critic, norm = _params(g, 'pre0/')
    flat_seg = {k: port.flatten_time_major(g[f'u0/segment/{k}']) for k in (
        'observations', 'actions', 'log_probs', 'returns', 'advantages')}
    g_actor, _ = port.clipped_ratio_grads(actor, flat_seg['observations'], flat_seg['actions'],
                                          flat_seg['advantages'], flat_seg['log_probs'])
    g_critic, _ = port.value_regression_grads(critic, norm[0], norm[1], flat_seg['observations'],
                                              flat_seg['returns'])
    keys = [k for k in before if 'normalizer' not in k]
    after = agent.model.state_dict()
    for key, grad in zip(keys, g_actor + g_critic):
        got = after[key].detach().cpu().numpy() - before[key]
        want = g['iter1/' + key] - before[key]
        live = np.abs(grad) > 1e-6 * np.abs(grad).max()     # see test_oracle_golden.py
        np.testing.assert_allclose(got[live], want[live], rtol=0, atol=1e-5, err_msg=key)


def test_agent_drop_in_trajectory(golden, lib):
    """The full drop-in path (agent.step / agent.update with NumPy in/out, Sequential
    collector, host-drawn noise) replays the reference's run: same actions, same stored
    segment, same update statistics."""
    import tonic_amd
    import tonic_amd.torch
    g = golden('ppo_halfcheetah_small')
    O, A, W, steps, seed, iterations, updates = (int(x) for x in g['cfg'])
    env = tonic_amd.environments.distribute(
        lambda: tonic_amd.environments.Synthetic(O, A, max_episode_steps=7), 1, W)
    env.initialize(seed=seed)
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations))
    agent.initialize(env.observation_space, env.action_space, seed=seed)
    observations = env.start()
    rng = np.random.RandomState(seed + 1)
    for t in range(steps):
        np.testing.assert_allclose(observations, g['act/observations'][t], rtol=0, atol=0)
        actions = agent.step(observations, t * W)
        np.testing.assert_allclose(actions, g['act/actions'][t], rtol=0, atol=3e-6)
        observations, infos = env.step(g['act/actions'][t])   # reference actions: same env path
        infos['rewards'] = (infos['rewards'] + rng.normal(size=W)).astype(np.float32)
        term = rng.uniform(size=W) < 0.05
        infos['terminations'] = term
        infos['resets'] = infos['resets'] | term
        agent.update(**infos, steps=t * W)
    infos = agent.last_infos
    n_actor = int(g['u0/info/actor/iterations'][0])
    assert (infos[0][:, 6] > 0).sum() == n_actor
    np.testing.assert_allclose(infos[0][:n_actor, 1], g['u0/info/actor/kl'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(infos[1][:, 0], g['u0/info/critic/loss'], rtol=1e-4, atol=1e-4)
    norm = agent.model.observation_normalizer
    np.testing.assert_allclose(norm._mean.detach().cpu().numpy(),
                               g['post0/observation_normalizer._mean'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(norm._std.detach().cpu().numpy(),
                               g['post0/observation_normalizer._std'], rtol=0, atol=1e-7)


@pytest.mark.parametrize('O,A,W', [(17, 6, 256), (3, 1, 5), (28, 8, 70)])
def test_fused_collect_step_equals_act_plus_store(lib, O, A, W):
    """tonic_ppo_collect_step == tonic_ppo_act + tonic_segment_store, bit for bit, and the
    hipGraph replay of the collect loop equals its eager execution."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    from tonic_amd.rollout import DeviceRollout
    T = 6
    results = []
    for fused, capture, packed in ((False, False, False), (True, False, False), (True, True, False),
                                   (True, False, True), (True, True, True)):
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T))
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=3)
        rollout = DeviceRollout(agent, W, T, seed=5, reset_probability=0.2, fused=fused,
                                packed=packed)
        rollout.collect(capture=capture)
        torch.cuda.synchronize()
        out = {k: agent.replay.buffers[k].cpu().numpy().copy() for k in (
            'observations', 'actions', 'next_observations', 'rewards', 'resets',
            'terminations', 'log_probs')}
        out['sums'] = agent.model.observation_normalizer.device_sums.cpu().numpy().copy()
        results.append(out)
    for other in results[1:3]:
        for key, want in results[0].items():
            assert np.array_equal(other[key], want), key
    # packed-weight 16x16x4 kernel: same mathematics, different float32 summation order
    assert all(np.array_equal(results[4][k], results[3][k]) for k in results[3])
    for key, want in results[0].items():
        if key in ('actions', 'log_probs'):
            np.testing.assert_allclose(results[3][key], want, rtol=0, atol=2e-5 if key == 'log_probs' else 3e-6)
        else:
            assert np.array_equal(results[3][key], want), key
    assert np.array_equal(results[0]['observations'], rollout.observations[:T].cpu().numpy())
    assert np.abs(results[0]['actions']).max() > 0 and np.isfinite(results[0]['log_probs']).all()


# ------------------------------------------------------------- drop-in plumbing end to end

def test_trainer_runs_ppo_end_to_end_and_checkpoints_interchange(lib, tmp_path):
    """`tonic_amd.Trainer` drives the PPO agent through the distributed collector like
    `python -m tonic.train` does (trainer.py:28-146): learner updates happen, the reference's log
    keys appear, and the `.pt` checkpoint has the reference's `state_dict` layout
    and restores the exact parameters."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments, logger

    def make_agent():
        return tonic_amd.torch.agents.PPO(
            replay=tonic_amd.replays.Segment(size=16, batch_iterations=3))

    logger.initialize(path=str(tmp_path))
    env = environments.distribute(lambda: environments.Synthetic(5, 2, max_episode_steps=7), 1, 4)
    env.initialize(seed=0)
    test_env = environments.distribute(lambda: environments.Synthetic(5, 2, max_episode_steps=7), 1, 1)
    test_env.initialize(seed=10000)
    agent = make_agent()
    agent.initialize(env.observation_space, env.action_space, seed=3)
    before = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    trainer = tonic_amd.Trainer(steps=160, epoch_steps=80, save_steps=160, show_progress=False)
    trainer.initialize(agent, env, test_env)
    trainer.run()
    header = open(tmp_path / 'log.csv').read().split('\n')[0].split(',')
    for key in ('actor/loss', 'actor/kl', 'actor/entropy', 'actor/clip_fraction', 'actor/std',
                'actor/stop', 'actor/iterations', 'critic/loss', 'critic/v', 'critic/iterations',
                'train/episode_score/mean', 'test/episode_score/mean', 'train/steps_per_second'):
        assert any(h == key or h.startswith(key + '/') for h in header), key
    after = agent.model.state_dict()
    assert any(not torch.equal(after[k].cpu(), before[k]) for k in before), 'no learner update ran'

    checkpoint = tmp_path / 'checkpoints' / 'step_160.pt'
    saved = torch.load(checkpoint, map_location='cpu')
    expected_keys = {
        'actor.torso.model.0.weight', 'actor.torso.model.0.bias', 'actor.torso.model.2.weight',
        'actor.torso.model.2.bias', 'actor.head.log_scale', 'actor.head.loc_layer.0.weight',
        'actor.head.loc_layer.0.bias', 'critic.torso.model.0.weight', 'critic.torso.model.0.bias',
        'critic.torso.model.2.weight', 'critic.torso.model.2.bias', 'critic.head.v_layer.weight',
        'critic.head.v_layer.bias', 'observation_normalizer._mean', 'observation_normalizer._std'}
    assert expected_keys <= set(saved)
    fresh = make_agent()
    fresh.initialize(env.observation_space, env.action_space, seed=99)
    fresh.load(str(checkpoint)[:-3])
    for key, value in fresh.model.state_dict().items():
        assert torch.equal(value.cpu(), after[key].cpu()), key
    # the flat device buffers the kernels read follow the loaded parameters
    flat = fresh.model.flat_actor.flat.cpu()
    first = fresh.model.state_dict()['actor.torso.model.0.weight'].cpu().reshape(-1)
    assert torch.equal(flat[:first.numel()], first)


@pytest.mark.parametrize('name,keys', [
    ('TRPO', ('actor/loss', 'actor/kl', 'actor/backtrack_steps', 'critic/loss', 'critic/v',
              'critic/iterations')),
    ('A2C', ('actor/loss', 'actor/kl', 'actor/entropy', 'actor/std', 'critic/loss', 'critic/v'))])
def test_trainer_runs_the_other_on_policy_agents(lib, tmp_path, name, keys):
    """A2C (a2c.py) and TRPO (trpo.py) through the same collector / Segment / Trainer path as PPO:
    updates happen, the parameters stay finite and the reference's log keys appear."""
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd import environments, logger
    logger.initialize(path=str(tmp_path))
    env = environments.distribute(lambda: environments.Synthetic(5, 2, max_episode_steps=7), 1, 4)
    env.initialize(seed=0)
    agent = getattr(tonic_amd.torch.agents, name)(
        replay=tonic_amd.replays.Segment(size=16, batch_iterations=3))
    agent.initialize(env.observation_space, env.action_space, seed=3)
    before = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    trainer = tonic_amd.Trainer(steps=160, epoch_steps=80, save_steps=160, show_progress=False)
    trainer.initialize(agent, env)
    trainer.run()
    header = open(tmp_path / 'log.csv').read().split('\n')[0].split(',')
    for key in keys:
        assert any(h == key or h.startswith(key + '/') for h in header), key
    after = agent.model.state_dict()
    assert all(torch.isfinite(v).all() for v in after.values())
    assert not torch.equal(after['actor.torso.model.0.weight'].cpu(),
                           before['actor.torso.model.0.weight'])
    assert not torch.equal(after['critic.torso.model.0.weight'].cpu(),
                           before['critic.torso.model.0.weight'])


def test_two_whole_iterations_at_baseline_size_vs_oracle(lib):
    """BASELINE cfg 2 at FULL size (T=4096 x W=256, N = 1 048 576): evaluate + GAE + two whole
    PPO iterations (actor grad -> reduce -> Adam, critic grad -> reduce -> Adam, twice) through
    the agent against oracle/torch_port.py (the reference's torch-CPU operators) — returns,
    losses, KL at 1e-5, parameter deltas after each iteration at 1e-5 on the elements whose
    gradient is above float32 summation noise.  Three times: with the critic's launches at the
    kernel's own width (256 workgroups), and at the widths they have in the mode bench.py
    measures — under the next rollout's resident collect kernel (`PPO._critic_width`): 219
    workgroups at 256 workers, 232 at 48 — i.e. tonic_value_regression_grad(max_workgroups =
    219 / 232) against the oracle at the metric's size."""
    import tonic_amd
    import tonic_amd.torch
    import torch_port
    from tonic_amd.environments import Box
    from tonic_amd.torch import updaters
    O, A, W, T = 17, 6, 256, 4096
    rng = np.random.RandomState(11)
    mean = (rng.standard_normal(O) * 0.1).astype(np.float32)
    std = (1 + 0.2 * rng.uniform(size=O)).astype(np.float32)
    observations = rng.standard_normal((T, W, O)).astype(np.float32)
    eps = rng.standard_normal((T, W, A)).astype(np.float32)
    resets = (rng.uniform(size=(T, W)) < 1e-3).astype(np.float32)
    data = dict(
        observations=observations,
        next_observations=rng.standard_normal((T, W, O)).astype(np.float32),
        rewards=rng.standard_normal((T, W)).astype(np.float32), resets=resets,
        terminations=resets * (rng.uniform(size=(T, W)) < 0.5).astype(np.float32))
    oracle_run = None

    for width in (0, 219, 232):
        agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=2))
        agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=2)
        agent._critic_width = lambda width=width: width
        state = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
        keys = [k for k in state if 'normalizer' not in k]
        if oracle_run is None:          # (same seed: every agent starts from these parameters)
            actor = [state[k] for k in ('actor.torso.model.0.weight', 'actor.torso.model.0.bias',
                                        'actor.torso.model.2.weight', 'actor.torso.model.2.bias',
                                        'actor.head.log_scale', 'actor.head.loc_layer.0.weight',
                                        'actor.head.loc_layer.0.bias')]
            critic = [state[k] for k in ('critic.torso.model.0.weight', 'critic.torso.model.0.bias',
                                         'critic.torso.model.2.weight', 'critic.torso.model.2.bias',
                                         'critic.head.v_layer.weight', 'critic.head.v_layer.bias')]
            actions, log_probs = port.ppo_act(actor, observations.reshape(-1, O), eps.reshape(-1, A))
            data['actions'], data['log_probs'] = actions.reshape(T, W, A), log_probs.reshape(T, W)
            oracle = torch_port.TorchPPO(O, A, steps=T)
            oracle.load(actor, critic, (mean, std))
            oracle.buffers = {k: v.copy() for k, v in data.items()}
            batch = oracle.evaluate_and_returns()
            want_infos, want_params, first_grads = [], [], None
            for it in range(2):
                a = oracle.actor_update(batch['observations'], batch['actions'], batch['advantages'],
                                        batch['log_probs'])
                c = oracle.critic_update(batch['observations'], batch['returns'])
                if first_grads is None:
                    first_grads = [p.grad.detach().numpy().copy()
                                   for p in oracle.actor_vars + oracle.critic_vars]
                want_infos.append((float(a['loss']), float(a['kl']), float(c['loss'])))
                want_params.append([p.detach().numpy().copy()
                                    for p in oracle.actor_vars + oracle.critic_vars])
            oracle_run = (state, want_infos, want_params, first_grads, oracle.buffers['returns'])
        first_state, want_infos, want_params, first_grads, want_returns = oracle_run
        assert all(np.array_equal(state[k], first_state[k]) for k in state)
        agent.model.observation_normalizer._mean.data.copy_(torch.as_tensor(mean))
        agent.model.observation_normalizer._std.data.copy_(torch.as_tensor(std))
        agent.replay._allocate(W, O, A)
        for key, value in data.items():
            agent.replay.buffers[key].copy_(torch.as_tensor(value))
        agent.replay.index = T

        # iteration by iteration on the device: one-iteration updates, parameters read in between
        agent.replay.batch_iterations = 1
        infos = agent.enqueue_update().cpu().numpy()
        assert agent.critic_updater.max_workgroups == width
        np.testing.assert_allclose(agent.replay.buffers['returns'].cpu().numpy(), want_returns,
                                   rtol=1e-5, atol=1e-5)
        got_infos = [(infos[0][0, 0], infos[0][0, 1], infos[1][0, 0])]
        got_params = [[agent.model.state_dict()[k].detach().cpu().numpy().copy() for k in keys]]
        # second iteration on the SAME returns / advantages (ppo.py:33-46 evaluates once per update)
        replay, actor_u, critic_u = agent.replay, agent.actor_updater, agent.critic_updater
        obs, act, raw_adv, old_lp, ret = next(iter(replay.learner_batches()))
        info2 = torch.zeros(2, updaters.INFO_WIDTH, device='cuda')
        actor_u.enqueue_grad(obs, act, raw_adv, replay.adv_stats, old_lp)
        critic_u.enqueue_grad(obs, ret)
        updaters.enqueue_step_pair(actor_u, critic_u, obs.shape[0], replay.adv_stats, info2[0], info2[1])
        info2 = info2.cpu().numpy()
        got_infos.append((info2[0, 0], info2[0, 1], info2[1, 0]))
        got_params.append([agent.model.state_dict()[k].detach().cpu().numpy().copy() for k in keys])
        np.testing.assert_allclose(np.array(got_infos), np.array(want_infos), rtol=1e-5, atol=1e-5,
                                   err_msg=f'critic width {width}')
        start = [state[k] for k in keys]
        for it in range(2):
            for key, first, got, want, grad in zip(keys, start, got_params[it], want_params[it],
                                                   first_grads):
                live = np.abs(grad) > 1e-6 * np.abs(grad).max()
                np.testing.assert_allclose(
                    (got - first)[live], (want - first)[live], rtol=0, atol=1e-5,
                    err_msg=f'critic width {width}, iteration {it + 1}: {key}')
        agent.close()


def test_value_regression_grad_width_is_an_argument(lib):
    """`max_workgroups` of tonic_value_regression_grad (tonic_ppo_actor_grad shares the code): the same width gives
    the same bits on every call, another width the same gradient sums at float32 rounding level
    (relative to the largest element), 0 = the kernel's own width, and a width above it changes
    nothing."""
    from tonic_amd import _lib
    O, A, n = 17, 6, 300000
    rng = np.random.RandomState(4)
    cparams = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
               rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
               rng.normal(size=(1, 64)) * 0.3, rng.normal(size=1)]
    obs = rng.standard_normal((n, O)).astype(np.float32)
    ret = rng.standard_normal(n).astype(np.float32)
    mean, std = np.zeros(O, np.float32), np.ones(O, np.float32)
    P = lib.tonic_v_critic_param_count(O)
    ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, A, 1), dtype=torch.uint8, device='cuda')
    keep = [dev(flat(cparams)), dev(mean), dev(std), dev(obs), dev(ret)]

    def critic(width):
        out = torch.zeros(P + 8, device='cuda')
        _lib.check(lib.tonic_value_regression_grad(
            *[t.data_ptr() for t in keep[:3]], 0.0, *[t.data_ptr() for t in keep[3:]],
            out.data_ptr(), n, O, width, ws.data_ptr(), ws.numel(), None), 'critic_grad')
        return out.cpu().numpy()

    full = critic(0)
    assert np.array_equal(critic(256), full) and np.array_equal(critic(4096), full)
    for width in (219, 232, 8, 1):
        got = critic(width)
        assert np.array_equal(critic(width), got), width
        assert not np.array_equal(got, full), 'another grouping of the partial sums'
        assert np.abs(got - full).max() <= 2e-6 * np.abs(full).max(), width
    out = torch.zeros(P + 8, device='cuda')
    rc = lib.tonic_value_regression_grad(
        *[t.data_ptr() for t in keep[:3]], 0.0, *[t.data_ptr() for t in keep[3:]],
        out.data_ptr(), n, O, -1, ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b'max_workgroups' in lib.tonic_last_error()      # TONIC_ERR_INVALID_ARGUMENT


def test_ppo_update_with_gradient_and_normaliser_clipping(golden, lib):
    """gradient_clip on both updaters (tonic_clip_grad_norm between the grad kernels and Adam;
    actors.py:96-98, critics.py:24-25) and MeanStd(clip=1.5) (the clamp in the critic kernels'
    input stage; mean_stds.py:37-38) against two consecutive updates of the reference."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    g = golden('ppo_clipped_small')
    O, A, W, steps, seed, iterations, updates = (int(x) for x in g['cfg'])
    actor_clip, critic_clip, normalizer_clip = (float(x) for x in g['clips'])
    model = tt.models.ActorCritic(
        actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(),
                              torso=tt.models.MLP((64, 64), torch.nn.Tanh),
                              head=tt.models.DetachedScaleGaussianPolicyHead()),
        critic=tt.models.Critic(encoder=tt.models.ObservationEncoder(),
                                torso=tt.models.MLP((64, 64), torch.nn.Tanh),
                                head=tt.models.ValueHead()),
        observation_normalizer=tt.normalizers.MeanStd(clip=normalizer_clip))
    agent = tt.agents.PPO(
        model=model, replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations),
        actor_updater=tt.updaters.ClippedRatio(gradient_clip=actor_clip),
        critic_updater=tt.updaters.VRegression(gradient_clip=critic_clip))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    for u in range(updates):
        state = {k[len(f'pre{u}/'):]: torch.as_tensor(g[k]) for k in g.files
                 if k.startswith(f'pre{u}/')}
        if u == 0:
            agent.model.load_state_dict(state)
        else:       # parameters continue from the first update; only the normaliser moved
            for key in ('observation_normalizer._mean', 'observation_normalizer._std'):
                target = dict(agent.model.state_dict())[key]
                target.copy_(state[key])
        before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
        agent.replay.index = 0
        _fill_segment(agent, g, u)
        infos = agent.enqueue_update().cpu().numpy()
        np.testing.assert_allclose(agent.replay.buffers['returns'].cpu().numpy(),
                                   g[f'u{u}/segment/returns'], rtol=1e-5, atol=1e-5)
        n_actor = int(g[f'u{u}/info/actor/iterations'][0])
        assert (infos[0][:, 6] > 0).sum() == n_actor
        np.testing.assert_allclose(infos[0][:n_actor, 1], g[f'u{u}/info/actor/kl'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(infos[0][:n_actor, 0], g[f'u{u}/info/actor/loss'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(infos[1][:, 0], g[f'u{u}/info/critic/loss'], rtol=2e-5, atol=1e-5)
        after = agent.model.state_dict()
        for key, start in before.items():
            if 'normalizer' in key:
                continue
            got = after[key].detach().cpu().numpy() - start
            want = g[f'post{u}/' + key] - start
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-5, err_msg=f'update {u}: {key}')
    report = agent.critic_updater.clip_workspace[-16:-8].view(torch.float32).cpu().numpy()
    assert 0 < report[0] < 1 and report[1] > critic_clip, 'the critic gradient must have been clipped'


def test_clip_grad_norm_against_torch(lib):
    """tonic_clip_grad_norm == torch.nn.utils.clip_grad_norm_ on the mean gradient, for norms
    above and below the bound, deterministic, and a no-op under the skip flag."""
    from tonic_amd import _lib
    rng = np.random.RandomState(5)
    for n, scale, max_norm in ((5708, 1.0 / 1048576, 0.05), (193538, 1.0 / 1024, 40.0), (7, 1.0, 100.0)):
        sums = (rng.standard_normal(n + 8) * (1.0 / scale) * 0.01).astype(np.float32)
        d = dev(sums)
        ws = torch.zeros(lib.tonic_clip_workspace_bytes(n), dtype=torch.uint8, device='cuda')
        _lib.check(lib.tonic_clip_grad_norm(d.data_ptr(), n, scale, max_norm, None, ws.data_ptr(),
                                            ws.numel(), None), 'clip')
        grad = torch.nn.Parameter(torch.zeros(n))
        grad.grad = torch.as_tensor(sums[:n]) * np.float32(scale)
        torch.nn.utils.clip_grad_norm_([grad], max_norm)
        got = d.cpu().numpy()
        np.testing.assert_allclose(got[:n] * np.float32(scale), grad.grad.numpy(), rtol=3e-6, atol=0)
        assert np.array_equal(got[n:], sums[n:]), 'the statistic slots are not gradients'
        d2 = dev(sums)
        _lib.check(lib.tonic_clip_grad_norm(d2.data_ptr(), n, scale, max_norm, None, ws.data_ptr(),
                                            ws.numel(), None), 'clip')
        assert torch.equal(d, d2)
        flag = torch.ones(1, dtype=torch.int32, device='cuda')
        d3 = dev(sums)
        _lib.check(lib.tonic_clip_grad_norm(d3.data_ptr(), n, scale, max_norm, flag.data_ptr(),
                                            ws.data_ptr(), ws.numel(), None), 'clip')
        assert np.array_equal(d3.cpu().numpy(), sums)


def test_a2c_update_matches_reference(golden, lib):
    """tonic_amd.torch.agents.A2C (StochasticPolicyGradient = the fused actor kernel in its plain
    mode + entropy bonus, then VRegression iterations) against two consecutive updates of the
    reference's A2C agent (a2c.py:101-127, actors.py:20-51)."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    g = golden('a2c_small')
    O, A, W, steps, seed, iterations, updates = (int(x) for x in g['cfg'])
    agent = tt.agents.A2C(
        replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations),
        actor_updater=tt.updaters.StochasticPolicyGradient(entropy_coeff=float(g['entropy_coeff'])))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    for u in range(updates):
        state = {k[len(f'pre{u}/'):]: torch.as_tensor(g[k]) for k in g.files
                 if k.startswith(f'pre{u}/')}
        if u == 0:
            agent.model.load_state_dict(state)
        else:
            for key in ('observation_normalizer._mean', 'observation_normalizer._std'):
                dict(agent.model.state_dict())[key].copy_(state[key])
        before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
        agent.replay.index = 0
        _fill_segment(agent, g, u)
        infos = agent.enqueue_update().cpu().numpy()
        for i, key in ((0, 'loss'), (1, 'kl'), (2, 'entropy'), (4, 'std')):
            np.testing.assert_allclose(infos[0, 0, i], g[f'u{u}/info/actor/{key}'][0], rtol=1e-5,
                                       atol=1e-5, err_msg=key)
        np.testing.assert_allclose(infos[1][:, 0], g[f'u{u}/info/critic/loss'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(infos[1][:, 1], g[f'u{u}/info/critic/v_mean'], rtol=1e-5, atol=1e-5)
        after = agent.model.state_dict()
        for key, start in before.items():
            if 'normalizer' in key:
                continue
            got = after[key].detach().cpu().numpy() - start
            np.testing.assert_allclose(got, g[f'post{u}/' + key] - start, rtol=0, atol=1e-5,
                                       err_msg=f'update {u}: {key}')


def test_trpo_update_matches_reference(golden, monkeypatch):
    """TRPO (trpo.py:7-97): evaluation, lambda-returns and the critic regression on the HIP engine,
    the actor step — conjugate gradient over autograd Fisher-vector products, backtracking line
    search (actors.py:115-156, optimizers.py:25-115) — as stock torch on the device, against two
    consecutive updates of the reference's TRPO agent.  The behaviour policy's locs / scales are
    recomputed from the stored observations instead of being stored per step."""
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    from tonic_amd.torch import agents as agents_module
    g = golden('trpo_small')
    O, A, W, steps, seed, iterations, updates = (int(x) for x in g['cfg'])
    agent = tt.agents.TRPO(replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    records = {}
    monkeypatch.setattr(agents_module.logger, 'store',
                        lambda key, value, stats=False: records.setdefault(key, []).append(
                            np.asarray(value)))
    # (the segment is filled without recording; the normaliser statistics come from the golden)
    monkeypatch.setattr(agent.model.observation_normalizer, 'update', lambda: None)
    for u in range(updates):
        state = {k[len(f'pre{u}/'):]: torch.as_tensor(g[k]) for k in g.files
                 if k.startswith(f'pre{u}/')}
        if u == 0:
            agent.model.load_state_dict(state)
        else:
            for key in ('observation_normalizer._mean', 'observation_normalizer._std'):
                dict(agent.model.state_dict())[key].copy_(state[key])
        before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
        agent.replay.index = 0
        _fill_segment(agent, g, u)
        records.clear()
        agent._update()
        for key in ('loss', 'kl'):
            np.testing.assert_allclose(records['actor/' + key][0], g[f'u{u}/info/actor/{key}'][0],
                                       rtol=2e-4, atol=1e-6, err_msg=key)
        assert int(records['actor/backtrack_steps'][0]) == int(g[f'u{u}/info/actor/backtrack_steps'][0])
        np.testing.assert_allclose(np.array(records['critic/loss']), g[f'u{u}/info/critic/loss'],
                                   rtol=1e-5, atol=1e-5)
        assert int(records['critic/iterations'][0]) == int(g[f'u{u}/info/critic/iterations'][0])
        after = agent.model.state_dict()
        for key, start in before.items():
            if 'normalizer' in key:
                continue
            got = after[key].detach().cpu().numpy() - start
            want = g[f'post{u}/' + key] - start
            # ten conjugate-gradient iterations in float32 amplify rounding differences between
            # the device's and NumPy's dot products: 1e-3 of the step, not 1e-5 of the parameter
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 + 2e-3 * np.abs(want).max(),
                                       err_msg=f'update {u}: {key}')


@pytest.mark.parametrize('O,A,n', [(111, 8, 20011), (376, 17, 4099), (40, 3, 777), (9, 21, 1500),
                                   (384, 32, 300)])
def test_wide_shapes_grads_vs_oracle(lib, O, A, n):
    """Shapes beyond the fused kernels (csrc/mlpwide.hip: layer-by-layer passes, activations in the
    workspace, weight gradients as per-slab partial images): gradients, statistics, both clipped
    branches, ragged row counts over many slabs, and MeanStd(clip) in the critic's first layer —
    against numpy_port at the tolerances of the fused kernels' tests."""
    rng = np.random.RandomState(O * 31 + A)
    actor = [(rng.standard_normal(s) * sc).astype(np.float32) for s, sc in (
        ((64, O), 0.3 / np.sqrt(O)), ((64,), 0.1), ((64, 64), 0.15), ((64,), 0.1), ((1, A), 0.3),
        ((A, 64), 0.15), ((A,), 0.1))]
    critic = [(rng.standard_normal(s) * sc).astype(np.float32) for s, sc in (
        ((64, O), 0.3 / np.sqrt(O)), ((64,), 0.1), ((64, 64), 0.15), ((64,), 0.1), ((1, 64), 0.2),
        ((1,), 0.1))]
    obs = rng.standard_normal((n, O)).astype(np.float32) * 1.5
    eps = rng.standard_normal((n, A)).astype(np.float32)
    actions, log_probs = port.ppo_act(actor, obs, eps)
    log_probs = (log_probs + rng.standard_normal(n) * 0.3).astype(np.float32)   # ratios off 1
    adv = rng.standard_normal(n).astype(np.float32)
    stats = np.array([0, 1, 0, 0], np.float32)
    got, P = actor_grad(lib, actor, obs, actions, adv, stats, log_probs)
    want, info = port.clipped_ratio_grads(actor, obs, actions, adv, log_probs)
    assert_grads_close(got[:P], want, n, 'wide actor grads')
    np.testing.assert_allclose(got[P + 0] / n, info['loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 1] / n, info['kl'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[P + 2] / n, info['clip_fraction'], atol=1e-7)
    np.testing.assert_allclose(got[P + 3] / n, info['entropy'], rtol=1e-5, atol=1e-5)
    assert got[P + 5] == n and 0.05 < info['clip_fraction'] < 0.95
    mean = (rng.standard_normal(O) * 0.2).astype(np.float32)
    std = (0.5 + rng.uniform(size=O)).astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    for clip in (0.0, 1.2):
        got_c, Pc = critic_grad(lib, critic, mean, std, obs, returns, clip=clip)
        want_c, info_c = port.value_regression_grads(critic, mean, std, obs, returns,
                                                     clip if clip > 0 else None)
        assert_grads_close(got_c[:Pc], want_c, n, f'wide critic grads (clip {clip})')
        np.testing.assert_allclose(got_c[Pc + 0] / n, info_c['loss'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got_c[Pc + 1] / n, info_c['v'].mean(), rtol=1e-5, atol=2e-6)
        assert got_c[Pc + 5] == n


@pytest.mark.parametrize('O,A', [(111, 8), (376, 17)])
def test_wide_shapes_agent_end_to_end(lib, O, A):
    """PPO on Ant-v3 / Humanoid-v3 shapes through the drop-in API (agent.step / agent.update with
    a Sequential environment, then one learner update) against the NumPy oracle."""
    import tonic_amd
    import tonic_amd.torch
    W, T, iterations = 6, 10, 4
    env = tonic_amd.environments.distribute(
        lambda: tonic_amd.environments.Synthetic(O, A, max_episode_steps=4), 1, W)
    env.initialize(seed=0)
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=T, batch_iterations=iterations))
    agent.initialize(env.observation_space, env.action_space, seed=0)
    state = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    observations = env.start()
    for t in range(T):
        actions = agent.step(observations, t * W)
        assert actions.shape == (W, A)
        observations, infos = env.step(actions)
        agent.update(**infos, steps=t * W)
    seg = {k: agent.replay.buffers[k].cpu().numpy() for k in (
        'observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
        'log_probs')}
    actor = [state[k] for k in ('actor.torso.model.0.weight', 'actor.torso.model.0.bias',
                                'actor.torso.model.2.weight', 'actor.torso.model.2.bias',
                                'actor.head.log_scale', 'actor.head.loc_layer.0.weight',
                                'actor.head.loc_layer.0.bias')]
    critic = [state[k] for k in ('critic.torso.model.0.weight', 'critic.torso.model.0.bias',
                                 'critic.torso.model.2.weight', 'critic.torso.model.2.bias',
                                 'critic.head.v_layer.weight', 'critic.head.v_layer.bias')]
    norm = (state['observation_normalizer._mean'], state['observation_normalizer._std'])
    new_actor, new_critic, infos, extra = port.ppo_update(actor, critic, norm, seg,
                                                          batch_iterations=iterations)
    assert seg['resets'].sum() > 0
    np.testing.assert_allclose(agent.replay.buffers['returns'].cpu().numpy(), extra['returns'],
                               rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    keys = [k for k in state if 'normalizer' not in k]
    for key, want in zip(keys, new_actor + new_critic):
        np.testing.assert_allclose(after[key].detach().cpu().numpy(), want, rtol=0, atol=2e-5,
                                   err_msg=key)
    np.testing.assert_allclose(agent.last_infos[1][:, 0],
                               [i['critic']['loss'] for i in infos], rtol=1e-5, atol=1e-5)


# ------------------------------------------------- torsos outside the hand-written kernels' shapes

def _generic_ppo_agent(g, steps, iterations):
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    O, A, seed = int(g['cfg'][0]), int(g['cfg'][1]), int(g['cfg'][4])
    sizes = tuple(int(x) for x in g['torso_sizes'])
    activation = getattr(torch.nn, str(g['torso_activation']))
    model = tt.models.ActorCritic(
        actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(),
                              torso=tt.models.MLP(sizes, activation),
                              head=tt.models.DetachedScaleGaussianPolicyHead()),
        critic=tt.models.Critic(encoder=tt.models.ObservationEncoder(),
                                torso=tt.models.MLP(sizes, activation), head=tt.models.ValueHead()),
        observation_normalizer=tt.normalizers.MeanStd())
    agent = tt.agents.PPO(model=model, replay=tonic_amd.replays.Segment(
        size=steps, batch_iterations=iterations))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    return agent


@pytest.mark.parametrize('path', ['hip', 'stock'])
@pytest.mark.parametrize('name', ['ppo_relu256_small', 'ppo_tanh3_small'])
def test_ppo_with_any_torso_matches_reference(golden, lib, name, path, monkeypatch):
    """models/utils.py:4-23 accepts any MLP(sizes, activation): PPO with MLP((256, 256), ReLU) and
    with three tanh layers (96, 48, 32) — torsos the fused kernels do not hold — replays the reference
    agent's run: identical initialisation from the seed, the acting trajectory (actions,
    log-probabilities), and one whole learner update (returns, per-iteration statistics, KL stop,
    parameter deltas).  `hip`: the layer-by-layer HIP path (the tonic_*_torso entries, csrc/mlpwide.hip:
    1 .. 4 layers of 4 .. 384 units, Tanh / ReLU); `stock`: what every torso outside that serves runs on —
    stock torch operators on the device (TONIC_AMD_TORSO_STOCK=1 forces it here)."""
    if path == 'stock':
        monkeypatch.setenv('TONIC_AMD_TORSO_STOCK', '1')
    g = golden(name)
    W, steps, iterations = int(g['cfg'][2]), int(g['cfg'][3]), int(g['cfg'][5])
    agent = _generic_ppo_agent(g, steps, iterations)
    for updater in (agent.actor_updater, agent.critic_updater):
        assert updater.stock == (path == 'stock') and (updater.torso is not None) == (path == 'hip')
    state = agent.model.state_dict()
    for key in state:       # identical initialisation from the same seed (CPU init parity)
        np.testing.assert_array_equal(state[key].cpu().numpy(), g['init/' + key], err_msg=key)
    # acting: the reference's observations and noise draws -> its actions and log-probabilities
    for t in range(3):
        eps = g['act/eps'][t]
        agent._randn = lambda workers, width, out=None, e=eps: torch.as_tensor(e)
        actions = agent.step(g['act/observations'][t], t * W)
        np.testing.assert_allclose(actions, g['act/actions'][t], rtol=0, atol=5e-6)
        np.testing.assert_allclose(agent._out.host_view('log_probs'), g['act/log_probs'][t],
                                   rtol=1e-5, atol=1e-5)
    # one whole update from the reference's segment and pre-update parameters
    agent = _generic_ppo_agent(g, steps, iterations)
    agent.model.load_state_dict({k[len('pre0/'):]: torch.as_tensor(g[k]) for k in g.files
                                 if k.startswith('pre0/')})
    before = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    _fill_segment(agent, g, 0)
    infos = agent.enqueue_update().cpu().numpy()
    b = agent.replay.buffers
    np.testing.assert_allclose(b['returns'].cpu().numpy(), g['u0/segment/returns'], rtol=1e-5, atol=1e-5)
    ran = infos[0][:, 6] > 0
    n_actor = int(g['u0/info/actor/iterations'][0])
    assert ran.sum() == n_actor and ran[:n_actor].all(), 'KL early stop'
    for i, key in enumerate(('loss', 'kl', 'entropy', 'clip_fraction', 'std')):
        np.testing.assert_allclose(infos[0][:n_actor, i], g[f'u0/info/actor/{key}'],
                                   rtol=1e-5, atol=1e-5, err_msg=key)
    assert np.array_equal(infos[0][:n_actor, 5] > 0.5, g['u0/info/actor/stop'])
    np.testing.assert_allclose(infos[1][:, 0], g['u0/info/critic/loss'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(infos[1][:, 1], g['u0/info/critic/v_mean'], rtol=1e-5, atol=1e-5)
    after = agent.model.state_dict()
    for key, start in before.items():
        if 'normalizer' in key:
            continue
        diff = np.abs((after[key].detach().cpu().numpy() - start) - (g['post0/' + key] - start))
        # (float32 summation order differs between torch-CPU and the device: a few Adam steps turn
        #  a gradient element at rounding level into a step of its own — DESIGN.md §2)
        assert (diff <= 2e-5).mean() >= 0.999 and diff.max() <= 2e-4, (key, diff.max())


@pytest.mark.parametrize('sizes,activation,O,A,n', [
    ((256, 256), 'ReLU', 17, 6, 20011), ((96, 48, 32), 'Tanh', 11, 3, 4099), ((128,), 'Tanh', 28, 8, 5000),
    ((64, 64, 64, 64), 'ReLU', 5, 2, 3000), ((384, 8), 'Tanh', 111, 8, 2500), ((400 - 16, 300), 'ReLU', 40, 21, 1037)])
def test_torso_grads_vs_float64_autograd(lib, sizes, activation, O, A, n):
    """The tonic_*_torso entries (any MLP(sizes, activation) of models/utils.py:4-23 on the layer-by-layer HIP
    path): gradient SUMS of the PPO actor loss and of the critic's squared error, the value forward and the
    acting forward against float64 torch autograd of the same networks — one to four layers, 4 .. 384 units,
    Tanh / ReLU, layers wider than one 64-output slice, ragged batches."""
    import ctypes
    from tonic_amd import _lib
    rng = np.random.RandomState(len(sizes) * 1000 + O)
    act = dict(Tanh=1, ReLU=2)[activation]
    fn = torch.tanh if act == 1 else torch.relu
    arr = (ctypes.c_int32 * len(sizes))(*sizes)
    dims = (O,) + tuple(sizes)
    torso = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        torso += [rng.normal(size=(fan_out, fan_in)) / np.sqrt(fan_in), rng.normal(size=fan_out) * 0.1]
    last = sizes[-1]
    actor = torso + [rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, last)) / np.sqrt(last),
                     rng.normal(size=A) * 0.1]
    critic = torso + [rng.normal(size=(1, last)) / np.sqrt(last), rng.normal(size=1)]
    actor, critic = [p.astype(np.float32) for p in actor], [p.astype(np.float32) for p in critic]
    assert lib.tonic_ppo_torso_param_count(O, A, 1, len(sizes), arr) == flat(actor).size
    assert lib.tonic_ppo_torso_param_count(O, 1, 0, len(sizes), arr) == flat(critic).size
    obs = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    returns = rng.standard_normal(n).astype(np.float32)
    mean = (rng.standard_normal(O) * 0.1).astype(np.float32)
    std = (1 + 0.3 * rng.uniform(size=O)).astype(np.float32)

    def f64(arrays):
        return [torch.tensor(np.asarray(a, np.float64), device='cuda', requires_grad=True) for a in arrays]

    def body(x, params):
        for W, b in zip(params[0::2], params[1::2]):
            x = fn(x @ W.T + b)
        return x
    x = torch.tensor(obs.astype(np.float64), device='cuda')
    pa = f64(actor)
    loc = torch.tanh(body(x, pa[:-3]) @ pa[-2].T + pa[-1])
    dist = torch.distributions.Normal(loc, (torch.nn.functional.softplus(pa[-3]) + 1e-8).clamp(1e-4, 1.0))
    a_t = torch.tensor(actions.astype(np.float64), device='cuda')
    old_lp = (dist.log_prob(a_t).sum(-1).detach().cpu().numpy() + rng.normal(size=n) * 0.1).astype(np.float32)
    ratio = torch.exp(dist.log_prob(a_t).sum(-1) - torch.tensor(old_lp.astype(np.float64), device='cuda'))
    adv_t = torch.tensor(adv.astype(np.float64), device='cuda')
    loss = -torch.min(adv_t * ratio, adv_t * ratio.clamp(0.8, 1.2)).sum()
    want_a = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, pa)]).cpu().numpy()
    pc = f64(critic)
    xn = (x - torch.tensor(mean.astype(np.float64), device='cuda')) / torch.tensor(std.astype(np.float64), device='cuda')
    v = (body(xn, pc[:-2]) @ pc[-2].T + pc[-1])[:, 0]
    want_c = torch.cat([g.reshape(-1) for g in torch.autograd.grad(
        ((v - torch.tensor(returns.astype(np.float64), device='cuda')) ** 2).sum(), pc)]).cpu().numpy()

    P, Pc = flat(actor).size, flat(critic).size
    ws = torch.empty(max(lib.tonic_ppo_torso_workspace_bytes(n, O, A, 1, len(sizes), arr),
                         lib.tonic_ppo_torso_workspace_bytes(n, O, 1, 0, len(sizes), arr)),
                     dtype=torch.uint8, device='cuda')
    keep = [dev(flat(actor)), dev(obs), dev(actions), dev(adv), dev(np.array([0, 1, 0, 0], np.float32)),
            dev(old_lp)]
    out = torch.zeros(P + 8, device='cuda')
    _lib.check(lib.tonic_ppo_actor_grad_torso(len(sizes), arr, act, *[t.data_ptr() for t in keep], out.data_ptr(),
                                              n, O, A, 0.2, 0.0, None, ws.data_ptr(), ws.numel(), None), 'actor')
    got_a = out.cpu().numpy()
    assert np.abs(got_a[:P] - want_a).max() <= 2e-5 * np.abs(want_a).max(), np.abs(got_a[:P] - want_a).max()
    np.testing.assert_allclose(got_a[P] / n, float(loss.detach()) / n, rtol=2e-5, atol=2e-6)
    keepc = [dev(flat(critic)), dev(mean), dev(std), dev(obs), dev(returns)]
    outc = torch.zeros(Pc + 8, device='cuda')
    _lib.check(lib.tonic_value_regression_grad_torso(
        len(sizes), arr, act, *[t.data_ptr() for t in keepc[:3]], 0.0, *[t.data_ptr() for t in keepc[3:]],
        outc.data_ptr(), n, O, ws.data_ptr(), ws.numel(), None), 'critic')
    got_c = outc.cpu().numpy()
    assert np.abs(got_c[:Pc] - want_c).max() <= 2e-5 * np.abs(want_c).max(), np.abs(got_c[:Pc] - want_c).max()
    values = torch.empty(n, device='cuda')
    _lib.check(lib.tonic_value_forward_torso(len(sizes), arr, act, *[t.data_ptr() for t in keepc[:3]], 0.0,
                                             keepc[3].data_ptr(), values.data_ptr(), n, O, ws.data_ptr(),
                                             ws.numel(), None), 'values')
    np.testing.assert_allclose(values.cpu().numpy(), v.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    eps = rng.standard_normal((n, A)).astype(np.float32)
    acts, lps = torch.empty(n, A, device='cuda'), torch.empty(n, device='cuda')
    keepe = dev(eps)
    _lib.check(lib.tonic_ppo_act_torso(len(sizes), arr, act, keep[0].data_ptr(), keep[1].data_ptr(),
                                       keepe.data_ptr(), acts.data_ptr(), lps.data_ptr(), n, O, A,
                                       ws.data_ptr(), ws.numel(), None), 'act')
    want_act = dist.loc.detach() + dist.scale.detach() * torch.tensor(eps.astype(np.float64), device='cuda')
    np.testing.assert_allclose(acts.cpu().numpy(), want_act.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lps.cpu().numpy(), dist.log_prob(want_act).sum(-1).detach().cpu().numpy(),
                               rtol=1e-5, atol=2e-5)
