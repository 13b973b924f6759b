"""CPU check of the fused grad kernel's lane/register index design (tests/mfma_emulator.py
mirrors tonic_amd/csrc/mlp64.hip) against the explicit back-propagation of the oracle."""
import numpy as np
import pytest

import mfma_emulator as emu
import mfma_emulator16 as emu16
import mfma_emulator16f as emu16f
import numpy_port as port


def _random_actor(rng, O, A):
    return [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
            rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
            rng.normal(size=(1, A)) * 0.3, rng.normal(size=(A, 64)) * 0.2,
            rng.normal(size=A) * 0.1]


EMULATORS = {'32x32x2': emu.emulate_grad, '16x16x4': emu16.emulate_grad16,
             'fp16x2, head on MFMA (shipped)': emu16f.emulate_grad16f}


@pytest.mark.parametrize('variant', list(EMULATORS))
@pytest.mark.parametrize('O,A,n', [(17, 6, 70), (3, 1, 33), (28, 8, 64), (20, 3, 40), (12, 8, 21)])
def test_actor_grad_layout(O, A, n, variant):
    rng = np.random.RandomState(O)
    params = [p.astype(np.float32) for p in _random_actor(rng, O, A)]
    obs = rng.normal(size=(n, O)).astype(np.float32)
    actions = np.clip(rng.normal(size=(n, A)), -1, 1).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    _, _, loc, scale, _ = port.ppo_actor_forward(params, obs)
    old_lp = port.normal_log_prob(actions, loc, scale) + rng.normal(size=n).astype(np.float32) * 0.3
    grads, stats = port.clipped_ratio_grads(params, obs, actions, adv, old_lp)
    data = dict(observations=obs, actions=actions, advantages=adv, log_probs=old_lp,
                clip=(np.float32(0.8), np.float32(1.2)))
    G, P = EMULATORS[variant]([p.astype(np.float64) for p in params], O, A, True, data, n)
    assert P == sum(g.size for g in grads)
    # the kernel leaves d loss / d sigma in the log_scale slot (chain rule in the reducer)
    _, dscale_dls = port.gaussian_scale(params[4])
    want = np.concatenate([g.reshape(-1) for g in grads]).astype(np.float64) * n
    got = G[:P].copy()
    o_ls = 64 * O + 64 + 4096 + 64
    got[o_ls:o_ls + A] *= dscale_dls.reshape(-1)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4 * np.abs(want).max())
    np.testing.assert_allclose(G[P + 0] / n, stats['loss'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(G[P + 1] / n, stats['kl'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(G[P + 2] / n, stats['clip_fraction'], atol=1e-6)
    assert G[P + 5] == n


@pytest.mark.parametrize('variant', list(EMULATORS))
@pytest.mark.parametrize('O,n', [(17, 70), (3, 5), (28, 96), (19, 37), (16, 16)])
def test_critic_grad_layout(O, n, variant):
    rng = np.random.RandomState(100 + O)
    params = [rng.normal(size=(64, O)) * 0.3, rng.normal(size=64) * 0.1,
              rng.normal(size=(64, 64)) * 0.15, rng.normal(size=64) * 0.1,
              rng.normal(size=(1, 64)) * 0.2, rng.normal(size=1)]
    params = [p.astype(np.float32) for p in params]
    mean = rng.normal(size=O).astype(np.float32)
    std = (np.abs(rng.normal(size=O)) + 0.5).astype(np.float32)
    obs = rng.normal(size=(n, O)).astype(np.float32)
    returns = rng.normal(size=n).astype(np.float32)
    grads, stats = port.value_regression_grads(params, mean, std, obs, returns)
    data = dict(observations=obs, returns=returns)
    G, P = EMULATORS[variant]([p.astype(np.float64) for p in params], O, 1, False, data, n,
                              norm=(mean.astype(np.float64), std.astype(np.float64)))
    want = np.concatenate([g.reshape(-1) for g in grads]).astype(np.float64) * n
    np.testing.assert_allclose(G[:P], want, rtol=2e-4, atol=2e-4 * np.abs(want).max())
    np.testing.assert_allclose(G[P + 0] / n, stats['loss'], rtol=1e-4)
    np.testing.assert_allclose(G[P + 1] / n, stats['v'].mean(), rtol=1e-4, atol=1e-5)
