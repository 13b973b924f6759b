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


def launch(world, out, port, command=(WORKER,), extra_env=None):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world),
               TONIC_AMD_BACKEND='gloo')
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, *command, out], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for p in procs:
        output = p.communicate(timeout=300)[0]
        assert p.returncode == 0, output[-3000:]


@pytest.mark.parametrize('schedule', ['joint', 'overlap'])
def test_two_ranks_equal_single_process(tmp_path, schedule):
    """Both exchange schedules (TONIC_AMD_EXCHANGE, agents.py PPO.enqueue_update)."""
    single, double = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    launch(1, single, 29631)
    launch(2, double, 29632, extra_env={'TONIC_AMD_EXCHANGE': schedule})
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


@pytest.mark.parametrize('kind,port', [('td3', 29641), ('sac', 29651), ('d4pg', 29661), ('mpo', 29871)])
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
        if kind == 'mpo':
            np.testing.assert_allclose(b['actor_infos'], a['actor_infos'], rtol=2e-4, atol=2e-5)
        for key in a.files:
            if key in ('infos', 'actor_infos'):
                continue
            # Adam turns a gradient element at float32-noise level into a +-lr step whose sign
            # depends on the summation order, so a handful of elements may
            # differ by a few lr; everything else agrees to 2e-5.
            diff = np.abs(b[key] - a[key])
            assert diff.max() < 3e-3 and np.mean(diff > 2e-5) < 1e-3, (key, diff.max())


@pytest.mark.parametrize('kind,port', [('td3', 29881), ('sac', 29891)])
def test_offpolicy_ranks_run_the_chained_launches_around_the_exchange(tmp_path, kind, port):
    """BASELINE config 4's path (TD3 sharded over the ranks; SAC alike): with several ranks the fused iteration
    runs in two halves around the gradient exchange (tonic_q_iteration_t.phase) instead of falling back to the
    split entry points — bit for bit their result at 2 and 4 ranks (TONIC_AMD_FUSED_PHASES=0 selects them)."""
    for world in (2, 4):
        outs = {}
        for phases in ('0', '1'):
            outs[phases] = str(tmp_path / f'{kind}{world}_{phases}.npz')
            launch(world, outs[phases], port + 2 * world + int(phases), command=(OFFPOLICY_WORKER, kind),
                   extra_env={'TONIC_AMD_FUSED_PHASES': phases})
        a, b = np.load(outs['0']), np.load(outs['1'])
        for key in a.files:
            assert np.array_equal(a[key], b[key]), (world, key, np.abs(a[key] - b[key]).max())


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


LOOP_WORKER = os.path.join(ROOT, 'tests', 'mp_ppo_loop_worker.py')


@pytest.mark.parametrize('exchange', ['rccl', 'oneshot'])
def test_two_ranks_keep_the_critic_under_the_next_rollout(tmp_path, exchange):
    """World size 2, host in the loop: the actor's iterations and THEIR all-reduces on the current
    stream, the critic's iterations and theirs on the second stream under the next rollout
    (agents.PPO._update) — the schedule of the 2 / 4 / 8-GPU points of the headline metric.  Against
    the same job with TONIC_AMD_CRITIC_OVERLAP=0 (the joint exchange of both networks' sums per
    iteration, everything on one stream): three rollouts + updates leave the same bits on both ranks
    — parameters, normaliser, every logged row (a two-rank sum does not depend on the order, and the
    critic's launches have the same width in both modes) — through the process group and through
    tonic_allreduce_f32."""
    outs = {}
    for overlap in ('1', '0'):
        outs[overlap] = str(tmp_path / f'loop{overlap}')
        launch(2, outs[overlap], 29760 + int(overlap) + (10 if exchange == 'oneshot' else 0),
               command=(LOOP_WORKER,),
               extra_env={'TONIC_AMD_CRITIC_OVERLAP': overlap, 'TONIC_AMD_ALLREDUCE': exchange})
    for rank in (0, 1):
        a = np.load(outs['1'] + f'.rank{rank}.npz')
        b = np.load(outs['0'] + f'.rank{rank}.npz')
        assert int(a['overlapped'][0]) == 3 and int(b['overlapped'][0]) == 0
        assert str(a['exchange']) == exchange
        for key in a.files:
            if key in ('overlapped', 'exchange'):
                continue
            assert np.array_equal(a[key], b[key]), (rank, key)
    first, second = (np.load(outs['1'] + f'.rank{r}.npz') for r in (0, 1))
    for key in first.files:                                  # replicas: identical across the ranks
        if key not in ('overlapped', 'exchange') and 'normalizer' not in key:
            assert np.array_equal(first[key], second[key]), key


ALLREDUCE_WORKER = os.path.join(ROOT, 'tests', 'mp_allreduce_worker.py')


@pytest.mark.parametrize('world', [2, 4, 8])
def test_one_shot_allreduce_is_the_rank_ordered_sum(tmp_path, world):
    """tonic_allreduce_f32 between `world` processes (sharing this box's GPU; windows exchanged as
    IPC handles): every rank ends with the float32 sum of the contributions taken in rank order —
    bit for bit, the same on all ranks — for buffer sizes of this path (11 k PPO, 178 k TD3 floats),
    odd tails, and 25 back-to-back calls without host synchronisation."""
    out = str(tmp_path / 'ar')
    launch(world, out, 29700 + world, command=(ALLREDUCE_WORKER,))
    ranks = [np.load(out + f'.rank{r}.npz') for r in range(world)]
    for call, n in enumerate((11101, 7, 177666, 4096, 11101, 11101, 1, 65536)):
        want = None
        for r in range(world):
            rng = np.random.RandomState(1000 * call + r)
            mine = (rng.standard_normal(n) * 10.0 ** rng.randint(-3, 4)).astype(np.float32)
            want = mine if want is None else (want + mine).astype(np.float32)
        for r in range(world):
            assert np.array_equal(ranks[r][f'call{call}'], want), (call, r)
    for r in range(1, world):
        assert np.array_equal(ranks[r]['chain'], ranks[0]['chain'])
    np.testing.assert_allclose(ranks[0]['chain'], (world + 1) / 2, rtol=1e-5)
    # the self-test parallel.one_shot runs before it selects this exchange passes here too; the
    # automatic selection itself declines on THIS box, unanimously and with the reason: the ranks
    # share a device (the windows are meant to be peer memory)
    for r in range(world):
        assert int(ranks[r]['self_test'][0]) == 1, str(ranks[r]['self_test_reason'])
        assert str(ranks[r]['choice']) == 'rccl' and int(ranks[r]['picked'][0]) == 0
        assert 'share devices' in str(ranks[r]['choice_reason'])


def test_one_shot_allreduce_between_peer_devices(tmp_path):
    """The same exchange with every rank on its OWN device (LOCAL_RANK -> cuda:r): the windows are
    peer memory reached over xGMI — the path tonic_allreduce_f32 exists for.  Needs a box with at
    least two GPUs (the 1-GPU boxes of the test tier skip it; the 8-GPU scaling node runs it)."""
    import torch
    devices = torch.cuda.device_count()
    if devices < 2:
        pytest.skip('one GPU on this box: peer windows need two')
    world = min(devices, 8)
    out = str(tmp_path / 'peer')
    launch(world, out, 29720 + world, command=(ALLREDUCE_WORKER,),
           extra_env={'TONIC_AMD_BACKEND': 'nccl'})
    ranks = [np.load(out + f'.rank{r}.npz') for r in range(world)]
    assert sorted(int(r['device']) for r in ranks) == list(range(world))
    for call, n in enumerate((11101, 7, 177666, 4096, 11101, 11101, 1, 65536)):
        want = None
        for r in range(world):
            rng = np.random.RandomState(1000 * call + r)
            mine = (rng.standard_normal(n) * 10.0 ** rng.randint(-3, 4)).astype(np.float32)
            want = mine if want is None else (want + mine).astype(np.float32)
        for r in range(world):
            assert np.array_equal(ranks[r][f'call{call}'], want), (call, r)
    for r in range(1, world):
        assert np.array_equal(ranks[r]['chain'], ranks[0]['chain'])
    # with a device per rank and peer access between every pair the learner's exchange selects the
    # one-shot all-reduce by itself (TONIC_AMD_ALLREDUCE=auto, set by the worker); anything else must come with a reason
    for r in range(world):
        assert int(ranks[r]['self_test'][0]) == 1, str(ranks[r]['self_test_reason'])
        assert str(ranks[r]['choice']) in ('oneshot', 'rccl')
        if str(ranks[r]['choice']) == 'oneshot':
            assert int(ranks[r]['picked'][0]) == 1
        else:
            assert str(ranks[r]['choice_reason']), 'a fallback to RCCL names its reason'
        print('rank', r, 'exchange:', ranks[r]['choice'], '-', ranks[r]['choice_reason'])


def test_two_ranks_with_the_one_shot_allreduce_equal_single_process(tmp_path):
    """The sharded PPO update with TONIC_AMD_ALLREDUCE=oneshot (tonic_allreduce_f32 instead of
    torch.distributed for the per-iteration gradient exchange) against the single-process update."""
    single, double = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    launch(1, single, 29731)
    launch(2, double, 29732, extra_env={'TONIC_AMD_ALLREDUCE': 'oneshot'})
    a, b = np.load(single), np.load(double)
    ran = a['infos'][0][:, 6] > 0
    assert np.array_equal(ran, b['infos'][0][:, 6] > 0)
    np.testing.assert_allclose(b['infos'][0][ran, :5], a['infos'][0][ran, :5], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b['infos'][1][:, :2], a['infos'][1][:, :2], rtol=1e-4, atol=1e-5)
    for key in a.files:
        if key in ('infos', 'adv_stats'):
            continue
        np.testing.assert_allclose(b[key], a[key], rtol=0, atol=2e-5, err_msg=key)


@pytest.mark.parametrize('worker,args,schedule', [(WORKER, (), 'joint'), (WORKER, (), 'overlap'),
                                                  (OFFPOLICY_WORKER, ('sac',), 'joint')])
def test_exchange_schedule_over_rccl_with_one_rank(tmp_path, worker, args, schedule):
    """The multi-rank learner schedule — all-reduces of the gradient sums (one joint per iteration, or
    one asynchronous per network hidden behind the other network's grad kernel), moments and
    normaliser sums — driven through the REAL RCCL
    backend ("nccl") with a process group of one rank (TONIC_AMD_EXERCISE_EXCHANGE=1): reductions
    are identities, so every output must equal the plain single-process run bit for bit."""
    plain, exchanged = str(tmp_path / 'plain.npz'), str(tmp_path / 'rccl.npz')
    launch(1, plain, 29761, command=(worker, *args))
    launch(1, exchanged, 29762, command=(worker, *args),
           extra_env={'TONIC_AMD_BACKEND': 'nccl', 'TONIC_AMD_EXERCISE_EXCHANGE': '1', 'RANK': '0',
                      'TONIC_AMD_EXCHANGE': schedule})
    a, b = np.load(plain), np.load(exchanged)
    for key in a.files:
        assert np.array_equal(a[key], b[key]), key


def _bench(*arguments, timeout=900):
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *arguments],
                         env=dict(os.environ), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [line for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher: bench.py starts the two ranks itself with the
    driver's launch line (python -m torch.distributed.run --nproc-per-node 2 ...; on this one-GPU
    box the ranks share the device under gloo and the line says so): ONE JSON line with n_gpus = 2,
    the replicated parameters bit-identical across ranks after the updates, and the strong-scaling
    leg (the metric's 256 workers split over the ranks) in the same ballpark as the weak one."""
    result = _bench('--gpus', '2', '--steps', '1', '--warmup', '1')
    assert result['n_gpus'] == 2 and result['scaling'] == 'weak'
    assert result['ranks_hold_identical_parameters'] is True
    assert result['config']['critic_under_next_rollout'] is True        # also with two ranks
    assert result['config']['global_workers'] == 512
    assert 'allreduce_us' in result and result['allreduce_us']['process_group_us'] > 0
    strong = result['strong_scaling']
    assert strong['global_workers'] == 256 and strong['workers_per_gpu'] == 128
    assert strong["ms_per_step"] < 3 * result["ms_per_step"]


def test_bench_refuses_a_rank_count_other_than_gpus():
    """A launcher environment that disagrees with --gpus must not produce a line claiming n_gpus."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps',
                          '1', '--warmup', '0', '--no-extras'], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode != 0 and 'n_gpus' in out.stderr
    assert not [line for line in out.stdout.splitlines() if line.startswith('{')]


@pytest.mark.parametrize('gpus', [1, 2])
def test_bench_cfg4_td3_sharded(gpus):
    """BASELINE config 4 through bench.py: TD3 humanoid-walk shapes, 512 workers sharded over the
    ranks, global index stream — one rank, and two ranks started by bench.py itself."""
    result = _bench('--workload', 'cfg4', '--gpus', str(gpus), '--steps', '2', '--warmup', '2',
                    '--no-extras')
    assert result['n_gpus'] == gpus and result['unit'] == 'updates/s' and result['value'] > 0
    assert result['scaling'] == 'strong'
    assert result['config']['workers_per_gpu'] == 512 // gpus
    assert result['allreduce_floats']['critics_every_iteration'] >= 177666 + 8   # (padded rows)
    if gpus > 1:
        assert result['ranks_hold_identical_parameters'] is True
        assert result['ranks_share_devices'] == 1 or result['rccl_ranks'] == gpus


LEARNING_WORKER = os.path.join(ROOT, 'tests', 'mp_learning_worker.py')


@pytest.mark.parametrize('case,port', [('PPO', 29811), ('TD3', 29821), ('SAC', 29831), ('TRPO', 29841),
                                       ('A2C', 29851), ('MPO', 29861)])
def test_two_ranks_learn_like_the_single_process_reference(tmp_path, case, port):
    """Whole training runs with one process per "GPU" (two ranks sharing this box's GPU, gloo):
    each rank steps HALF of the workers through the Trainer — rank-offset environment seeds, the
    rows of its own workers out of the global noise draws (TONIC_AMD_GLOBAL_NOISE=1), its shard of
    Segment / Buffer, all-reduced gradients, moments and normaliser sums, global step counting.
    Together the ranks are one run with all the workers: the mean of their reward curves must be
    the curve of the UNMODIFIED single-process reference (tests/golden/learning_curves.json), and
    the replicated parameters must be identical on both ranks."""
    import json
    out = str(tmp_path / 'run')
    launch(2, out, port, command=(LEARNING_WORKER, case),
           extra_env={'TONIC_AMD_GLOBAL_NOISE': '1'})
    ranks = [json.load(open(f'{out}.rank{r}.json')) for r in range(2)]
    assert ranks[0]['parameters'] == ranks[1]['parameters']
    curve = np.mean([r['curve'] for r in ranks], axis=0)
    golden = os.path.join(ROOT, 'tests', 'golden', 'learning_curves.json')
    reference = np.array(json.load(open(golden))['curves'][case])
    print('LEARN 2 ranks', case, ' '.join(f'{x:.3f}' for x in curve))
    np.testing.assert_allclose(curve, reference, rtol=0, atol=0.01, err_msg=case)
