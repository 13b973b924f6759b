"""The N-rank learner path on one GPU box: two processes (gloo all-reduce of the flat gradient-sum
buffers, advantage moments and normaliser sums; both ranks share cuda:0) must reproduce the
single-process full-batch PPO update — the property the RCCL run over 2/4/8 GPUs relies on."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'mp_ppo_worker.py')


OFFPOLICY_WORKER = os.path.join(ROOT, 'tests', 'mp_offpolicy_worker.py')


def launch(world, out, port, command=(WORKER,)):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world),
               TONIC_AMD_BACKEND='gloo')
    procs = [subprocess.Popen([sys.executable, *command, out], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for p in procs:
        output = p.communicate(timeout=300)[0]
        assert p.returncode == 0, output[-3000:]


def test_two_ranks_equal_single_process(tmp_path):
    single, double = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    launch(1, single, 29631)
    launch(2, double, 29632)
    a, b = np.load(single), np.load(double)
    np.testing.assert_allclose(b['adv_stats'], a['adv_stats'], rtol=1e-5, atol=1e-6)
    ran = a['infos'][0][:, 6] > 0
    assert np.array_equal(ran, b['infos'][0][:, 6] > 0)
    np.testing.assert_allclose(b['infos'][0][ran, :5], a['infos'][0][ran, :5], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b['infos'][1][:, :2], a['infos'][1][:, :2], rtol=1e-4, atol=1e-5)
    for key in a.files:
        if key in ('infos', 'adv_stats'):
            continue
        np.testing.assert_allclose(b[key], a[key], rtol=0, atol=2e-5, err_msg=key)


@pytest.mark.parametrize('kind,port', [('td3', 29641), ('sac', 29651)])
def test_offpolicy_two_ranks_equal_single_process(tmp_path, kind, port):
    """Sharded Buffer + global index / noise streams: 1, 2 and 4 ranks give the same
    TD3 / SAC update as one process holding the whole buffer."""
    outs = {}
    for world in (1, 2, 4):
        outs[world] = str(tmp_path / f'{kind}{world}.npz')
        launch(world, outs[world], port + world, command=(OFFPOLICY_WORKER, kind))
    a = np.load(outs[1])
    for world in (2, 4):
        b = np.load(outs[world])
        np.testing.assert_allclose(b['infos'], a['infos'], rtol=1e-4, atol=1e-5)
        for key in a.files:
            if key == 'infos':
                continue
            # Adam turns a gradient element at float32-noise level into a +-lr step whose sign
            # depends on the summation order, so a handful of elements may
            # differ by a few lr; everything else agrees to 2e-5.
            diff = np.abs(b[key] - a[key])
            assert diff.max() < 3e-3 and np.mean(diff > 2e-5) < 1e-3, (key, diff.max())


def test_buffer_get_yields_each_ranks_part_of_the_global_batch(tmp_path):
    """Buffer.get with 1 / 2 / 4 ranks: every rank yields exactly its rows of each globally drawn
    batch (possibly none, never an uninitialised tail); the union over ranks is the batch one
    process holding the whole buffer yields."""
    outs = {}
    for world in (1, 2, 4):
        outs[world] = str(tmp_path / f'get{world}.npz')
        launch(world, outs[world], 29670 + world, command=(OFFPOLICY_WORKER, 'td3'))
    whole = np.load(outs[1] + '.get0.npz')
    iterations = len([k for k in whole.files if k.startswith('rewards')])
    for world in (2, 4):
        parts = [np.load(outs[world] + f'.get{r}.npz') for r in range(world)]
        empty = 0
        for i in range(iterations):
            rewards = np.concatenate([p[f'rewards{i}'] for p in parts])
            observations = np.concatenate([p[f'observations{i}'] for p in parts])
            assert rewards.shape == whole[f'rewards{i}'].shape
            assert np.isfinite(observations).all()
            order, want = np.argsort(rewards), np.argsort(whole[f'rewards{i}'])
            assert np.array_equal(rewards[order], whole[f'rewards{i}'][want])
            assert np.array_equal(observations[order], whole[f'observations{i}'][want])
            empty += sum(p[f'rewards{i}'].shape[0] == 0 for p in parts)
        assert empty > 0, 'the case of a rank drawing nothing must be exercised'
