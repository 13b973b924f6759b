"""CPU tests of the host side: collector semantics vs the reference goldens, the C-ABI
library's exported symbols vs include/tonic_hip.h, trainer bookkeeping, sharding helpers and
the world_size-2 gloo path of the gradient-sum exchange."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import numpy_port as port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tonic_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'tonic_hip.h')).read()
    developer = open(os.path.join(ROOT, 'include', 'tonic_hip_dev.h')).read()
    public = set(re.findall(r'\b(tonic_[a-z0-9_]+)\s*\(', header)) - {'tonic_status'}
    # the developer entries (tuning switch, stand-alone GEMM, cycle probe) have their own header
    assert not public & {'tonic_set_tuning', 'tonic_gemm_f32', 'tonic_debug_grad16_phases'}
    declared = public | set(re.findall(r'\b(tonic_[a-z0-9_]+)\s*\(', developer))
    assert {'tonic_gae_lambda_returns', 'tonic_ppo_actor_grad', 'tonic_adam_step'} <= declared
    assert declared == set(_lib.SIGNATURES), 'ctypes table and header disagree'
    lib = ctypes.CDLL(_lib.LIBRARY_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} missing from libtonic_hip.so'
    # ... and every prototype has as many parameters as its ctypes entry passes (an argument added on
    # one side only would still load and then read garbage)
    counted = 0
    for text in (header, developer):
        for name, params in re.findall(r'\b(tonic_[a-z0-9_]+)\s*\(([^;{}()]*)\)\s*;', text):
            params = re.sub(r'/\*.*?\*/', '', params, flags=re.S).strip()
            count = 0 if params in ('', 'void') else params.count(',') + 1
            assert count == len(_lib.SIGNATURES[name][1]), \
                f'{name}: {count} parameters in the header, {len(_lib.SIGNATURES[name][1])} in _lib.py'
            counted += 1
    assert counted >= len(declared) - 2, (counted, len(declared))
    loaded = _lib.load()
    assert loaded.tonic_abi_version() == _lib.ABI_VERSION
    assert loaded.tonic_target_arch() == b'gfx950'
    assert loaded.tonic_ppo_actor_param_count(17, 6) == 5708
    assert loaded.tonic_v_critic_param_count(17) == 5377
    # argument validation happens before any GPU work: error codes + messages, no crash
    assert loaded.tonic_gae_workspace_bytes(4096, 256, 0) > 0
    status = loaded.tonic_ppo_act(None, None, None, None, None, 4, 17, 6, None)
    assert status == -1 and b'null' in loaded.tonic_last_error()
    status = loaded.tonic_set_tuning(b'no_such_knob', 1)
    assert status == -1 and b'unknown key' in loaded.tonic_last_error()


def test_offpolicy_size_queries_and_hidden_codes():
    """Host-side queries of the off-policy entries (no GPU work): the `H` argument carries a plain width of any size
    as it is and packs every other two-layer torso with bit 30 set (plain widths of 1 024 and more used to
    fall off the kernels); the workspaces hold the fp16x2 weight images of the networks (csrc/mlpimg.h: [16-feature
    tiles][32-wide k-chunks][hi, lo] blocks of 1 KB per term) where the image passes serve the shape, and none where
    they do not."""
    from tonic_amd import _lib
    lib = _lib.load()
    for width in (16, 256, 1023, 1024, 4000):
        assert lib.tonic_mlp_hidden(width, width, 1) == width
    packed = lib.tonic_mlp_hidden(400, 300, 1)
    assert packed & (1 << 30) and packed & 4095 == 400 and (packed >> 12) & 4095 == 300 and (packed >> 24) & 7 == 1
    assert lib.tonic_mlp_hidden(256, 256, 2) & (1 << 30)          # Tanh: not the plain torso
    assert lib.tonic_mlp_hidden(4096, 256, 1) == -1 and lib.tonic_mlp_hidden(256, 256, 4) == -1

    def image_bytes(M, K):
        return ((M + 15) // 16) * ((K + 31) // 32) * 2 * 1024
    O, A, H = 111, 8, 256
    # SAC actor: W1, W2, W2^T, and per head Wh, Wh^T
    want = image_bytes(H, O) + 2 * image_bytes(H, H) + 2 * (image_bytes(A, H) + image_bytes(H, A))
    assert lib.tonic_mlp_actor_image_bytes(O, H, A, 2) == (want + 255) // 256 * 256
    assert lib.tonic_mlp_actor_image_bytes(O, H, A, 1) < lib.tonic_mlp_actor_image_bytes(O, H, A, 2)
    assert lib.tonic_mlp_actor_image_bytes(O, 400, A, 1) == 0      # wider than the fused passes hold
    assert lib.tonic_mlp_actor_image_bytes(O, packed, A, 1) == 0   # not a plain torso
    critic = image_bytes(H, O + A) + 2 * image_bytes(H, H) + image_bytes(A, H)
    images = 2 * lib.tonic_mlp_actor_image_bytes(O, H, A, 2) + 2 * ((2 * critic + 255) // 256 * 256)
    assert lib.tonic_q_iteration_workspace_bytes(1024, O, A, H) >= images
    assert lib.tonic_offpolicy_workspace_bytes(1024, O, A, H) >= images
    assert lib.tonic_q_iteration_supported(O, H, A, 2) == 1 and lib.tonic_q_iteration_supported(O, packed, A, 2) == 0

    # the next iteration's policy passes riding in a critic-step launch (tonic_q_iteration_t.ahead): only while every
    # workgroup of the launch is resident at once — tiles x (2 critics x 2 + passes) + 1 <= 256
    assert lib.tonic_q_iteration_ahead_supported(100, 67, 256, 21, 2, 2) == 1        # cfg 4: 7 x 6 + 1
    assert lib.tonic_q_iteration_ahead_supported(672, 67, 256, 21, 2, 2) == 1        # 42 x 6 + 1 = 253
    assert lib.tonic_q_iteration_ahead_supported(673, 67, 256, 21, 2, 2) == 0
    assert lib.tonic_q_iteration_ahead_supported(1024, 111, 256, 8, 2, 1) == 0       # cfg 3's batch: 64 x 5 + 1
    assert lib.tonic_q_iteration_ahead_supported(100, 67, packed, 21, 2, 2) == 0     # not on the image passes
    assert lib.tonic_q_iteration_ahead_supported(100, 67, 256, 21, 2, 3) == 0
    # ... and argument errors come back before anything touches a device: a stage other than 0 / 2, a third slot,
    # passes ahead of an iteration that steps the actor itself
    import ctypes                                   # (TONIC_ERR_INVALID_ARGUMENT = -1)
    def call(**fields):
        args = _lib.QIteration(kind=0, actor_due=0, B=100, O=67, H=256, A=21, **fields)
        return lib.tonic_q_iteration(ctypes.byref(args), None)
    assert call(stage=1) == -1 and 'stage 1' in lib.tonic_last_error().decode()
    assert call(slot=2) == -1
    follower = _lib.QIteration(kind=0, actor_due=1, B=100, O=67, H=256, A=21, slot=1)
    args = _lib.QIteration(kind=0, actor_due=1, B=100, O=67, H=256, A=21, ahead=ctypes.addressof(follower))
    assert lib.tonic_q_iteration(ctypes.byref(args), None) == -1
    assert 'policy passes ahead' in lib.tonic_last_error().decode()


def test_agents_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import tonic_amd.torch
    from tonic_amd import _lib
    from tonic_amd.environments import Box
    agent = tonic_amd.torch.agents.PPO()
    with pytest.raises(_lib.TonicHipError, match='no CPU fallback'):
        agent.initialize(Box(-1, 1, (3,)), Box(-1, 1, (1,)), seed=0)


def test_sequential_matches_reference_golden(golden):
    from tonic_amd import environments
    g = golden('sequential')
    env = environments.distribute(
        lambda: environments.Synthetic(3, 2, max_episode_steps=int(g['max_episode_steps'])), 1, 4)
    env.initialize(seed=int(g['seed']))
    obs = env.start()
    assert obs.dtype == np.float32 and np.array_equal(obs, g['observations'][0])
    for t in range(g['actions'].shape[0]):
        obs, infos = env.step(g['actions'][t])
        assert np.array_equal(obs, g['observations'][t + 1])
        assert np.array_equal(infos['observations'], g['next_observations'][t])
        assert np.array_equal(infos['rewards'], g['rewards'][t])
        assert infos['resets'].dtype == bool and np.array_equal(infos['resets'], g['resets'][t])
        assert np.array_equal(infos['terminations'], g['terminations'][t])


def test_parallel_equals_sequential():
    """Sequential(P*S) == Parallel(P, S) for equal seeds (seed layout distributed.py:18-20,109)."""
    from tonic_amd import environments

    def builder():
        return environments.Synthetic(5, 3, max_episode_steps=4)
    seq = environments.distribute(builder, 1, 6)
    par = environments.distribute(builder, 3, 2)
    assert isinstance(par, environments.Parallel)
    seq.initialize(seed=11)
    par.initialize(seed=11)
    a, b = seq.start(), par.start()
    assert np.array_equal(a, b)
    rng = np.random.RandomState(0)
    for _ in range(9):
        actions = rng.uniform(-1, 1, size=(6, 3)).astype(np.float32)
        (oa, ia), (ob, ib) = seq.step(actions), par.step(actions)
        assert np.array_equal(oa, ob)
        for key in ia:
            assert np.array_equal(ia[key], ib[key]), key


def test_step_outputs_are_read_only_views_or_copies():
    """The zero-copy step outputs alias the shared block (the next step and the GPU overwrite
    them in place): they are handed out read-only, the same objects every step; with
    copy_outputs=True the caller gets fresh writable arrays like the reference returns."""
    from tonic_amd import environments

    def builder():
        return environments.Synthetic(5, 3, max_episode_steps=4)
    env = environments.distribute(builder, 1, 3)
    env.initialize(seed=1)
    first = env.start()
    obs, infos = env.step(np.zeros((3, 3), np.float32))
    assert obs is first and not obs.flags.writeable
    with pytest.raises(ValueError):
        obs[0, 0] = 1.0
    for value in infos.values():
        assert not value.flags.writeable
    again, infos2 = env.step(np.zeros((3, 3), np.float32))
    assert again is obs and all(infos2[k] is infos[k] for k in infos)
    copying = environments.distribute(builder, 1, 3, copy_outputs=True)
    copying.initialize(seed=1)
    kept = copying.start()
    obs, infos = copying.step(np.zeros((3, 3), np.float32))
    assert obs is not kept and obs.flags.writeable and infos['rewards'].flags.writeable
    before = kept.copy()
    copying.step(np.zeros((3, 3), np.float32))
    assert np.array_equal(kept, before)


def test_noise_drawn_ahead_is_the_reference_stream():
    """_NoiseAhead (the collect loop's action noise, drawn 64 steps per call by a helper thread)
    hands out exactly the draws of consecutive torch.randn(W, A) calls, and `rewind` leaves the
    generator where those calls would have left it — with and without an unconsumed draw, across
    buffer swaps; shapes the block draw cannot reproduce fall back to one draw per step."""
    import torch
    from tonic_amd.torch.agents import _NoiseAhead

    class Agent:
        global_noise, _acting_generator = False, None

        def _randn(self, workers, width, out=None):
            return torch.randn(workers, width, out=out)
    for workers, width, bulk in ((256, 6, True), (16, 1, True), (8, 6, True), (5, 3, False)):
        torch.manual_seed(3)
        reference = [torch.randn(workers, width).numpy().copy() for _ in range(200)]
        after = torch.randn(4).numpy().copy()
        torch.manual_seed(3)
        agent = Agent()                      # (kept alive: the helper thread ends with its agent)
        noise = _NoiseAhead(agent, workers, width)
        assert noise.bulk == bulk
        out = np.zeros((workers, width), np.float32)
        for t in range(150):
            noise.take(out)
            assert np.array_equal(out, reference[t]), (workers, width, t)
        noise.rewind(1)                      # the 150th draw has not been consumed: un-draw it
        assert np.array_equal(torch.randn(workers, width).numpy(), reference[149])
        for t in range(150, 200):
            noise.take(out)
            assert np.array_equal(out, reference[t]), (workers, width, t)
        noise.rewind(0)
        assert np.array_equal(torch.randn(4).numpy(), after)
        # a helper whose agent is gone (closed) draws nothing ahead: same stream, drawn in place,
        # also when the close comes while a buffer drawn ahead is pending
        torch.manual_seed(3)
        agent = Agent()
        noise = _NoiseAhead(agent, workers, width)
        for t in range(200):
            if t == 70:
                noise.close()
            noise.take(out)
            assert np.array_equal(out, reference[t]), (workers, width, t, 'closed')
        noise.rewind(0)
        assert np.array_equal(torch.randn(4).numpy(), after)


def test_synthetic_batch_protocol():
    from tonic_amd.environments import SyntheticBatch
    env = SyntheticBatch(8, 17, 6, max_episode_steps=3, termination_probability=0.2)
    env.initialize(seed=0)
    obs = env.start()
    assert obs.shape == (8, 17) and obs.dtype == np.float32
    for t in range(7):
        obs, infos = env.step(np.zeros((8, 6), np.float32))
        assert (infos['resets'] | ~infos['terminations']).all()      # resets ⊇ terminations
        assert infos['rewards'].dtype == np.float32
        keep = ~infos['resets']
        assert np.array_equal(obs[keep], infos['observations'][keep])


def test_environments_ring_the_block_when_their_step_record_is_complete():
    """tonic_collector_ring / the `ring` argument of tonic_collector_synthetic_step / the last
    worker group's tonic_collector_worker_done issue the command an agent has ARMED in the block's
    header (tonic_collector_arm, GPU side) — host stores only.  Without a GPU the arming side is
    played by hand: a word in the header's `armed` field must move to `command` exactly once, at the
    step whose record it waits for, and nothing happens when nothing is armed."""
    from tonic_amd import _lib, environments
    lib = _lib.load()
    O, A, W = 5, 2, 6

    def words(block):
        # (BlockHeader, csrc/collector.hip: 64-byte aligned words behind the offsets table)
        return np.frombuffer(block.memory, np.uint64, 512, 0)

    for kind in ('batch', 'sequential', 'parallel'):
        if kind == 'batch':
            env = environments.SyntheticBatch(W, O, A, max_episode_steps=4, pool=3)
        else:
            env = environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=4),
                                          2 if kind == 'parallel' else 1, W // 2 if kind == 'parallel' else W)
        env.initialize(seed=1)
        env.start()
        block = env.block
        head = words(block)
        assert lib.tonic_collector_ring(block.address) == 0          # nothing armed
        # find the two words: arm a recognisable pattern through every 64-byte slot that is zero and
        # see which one a ring moves (keeps this test independent of the header's exact layout)
        armed_at = command_at = None
        for slot in range(16, 512, 8):
            if head[slot] != 0:
                continue
            before = head.copy()
            head[slot] = (7 << 32) | 0x106
            if lib.tonic_collector_ring(block.address) == 1:
                changed = [i for i in range(512) if head[i] != before[i]]
                armed_at = slot
                command_at = [i for i in changed if head[i] == (7 << 32) | 0x106][0]
                break
            head[slot] = 0
        assert armed_at is not None and head[armed_at] == 0, kind
        # an environment step rings exactly once, whoever completes the record
        for t in range(6):
            head[command_at] = 0
            head[armed_at] = ((8 + t) << 32) | 0x206
            actions = block.out_actions if kind == 'batch' else np.zeros((W, A), np.float32)
            env.step(actions)
            assert head[armed_at] == 0 and head[command_at] == ((8 + t) << 32) | 0x206, (kind, t)
            head[command_at] = 0
            assert lib.tonic_collector_ring(block.address) == 0
            assert head[command_at] == 0
        if hasattr(env, 'close'):
            env.close()


def test_push_protocol_of_the_block_header_on_ordinary_memory():
    """Collector transport 3 without a GPU: the header's `push_window` / `push_pid` (csrc/collector.hip) name a
    window the OWNING process stores the step's observation rows and the command word into.  Here the window is
    ordinary host memory: a ring in the owning process must put the rows there BEFORE the command and make the
    block's own copy as well; with another owner recorded (what a forked worker group sees) the armed command must
    stay where it is — for all three environments, the workers of `Parallel` included."""
    import os
    from tonic_amd import _lib, environments
    lib = _lib.load()
    O, A, W = 5, 2, 6
    for kind in ('batch', 'sequential', 'parallel'):
        if kind == 'batch':
            env = environments.SyntheticBatch(W, O, A, max_episode_steps=50, pool=3)
        else:
            env = environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=50),
                                          2 if kind == 'parallel' else 1, W // 2 if kind == 'parallel' else W)
        env.initialize(seed=1)
        env.start()
        block = env.block
        head = np.frombuffer(block.memory, np.uint64, 512, 0)
        head32 = np.frombuffer(block.memory, np.int32, 1024, 0)
        # the `armed` and `command` words, found as test_environments_ring_the_block... finds them
        armed_at = command_at = None
        for slot in range(16, 512, 8):
            if head[slot] != 0:
                continue
            before = head.copy()
            head[slot] = (7 << 32) | 0x106
            if lib.tonic_collector_ring(block.address) == 1:
                armed_at = slot
                command_at = [i for i in range(512) if head[i] != before[i] and head[i] == (7 << 32) | 0x106][0]
                break
            head[slot] = 0
        head[command_at] = 0
        window_at, pid_at = armed_at + 8, 2 * (armed_at + 8) + 2       # the next 64-byte line: u64 window, i32 pid
        window = np.zeros(block.nbytes, np.uint8)
        window_words = window.view(np.uint64)
        obs_offset = lib.tonic_collector_block_offset(block.address, 1)
        rows = lambda memory: np.frombuffer(memory, np.float32, W * O, obs_offset).reshape(W, O)
        actions = block.out_actions if kind == 'batch' else np.zeros((W, A), np.float32)
        # (1) this process owns the window
        head[window_at] = window.ctypes.data
        head32[pid_at] = os.getpid()
        for t in range(3):
            word = ((8 + t) << 32) | 0x206
            head[armed_at] = word
            observations, _ = env.step(actions)
            assert head[armed_at] == 0 and head[command_at] == word, (kind, t)
            assert window_words[command_at] == word, (kind, t)
            assert np.array_equal(rows(window), block.observations), (kind, t)
            assert np.array_equal(np.asarray(observations), block.observations), (kind, t)
            assert block.observations.any()
        # (2) somebody else's window: nothing may be issued from here, nothing stored through the address
        head32[pid_at] = os.getpid() + 1
        window[:] = 0
        head[command_at] = 0
        head[armed_at] = (20 << 32) | 0x206
        env.step(actions)
        assert head[armed_at] == (20 << 32) | 0x206 and head[command_at] == 0, kind
        assert lib.tonic_collector_ring(block.address) == 0 and not window.any(), kind
        head[armed_at] = 0
        head[window_at] = 0
        if hasattr(env, 'close'):
            env.close()


def test_vectorcall_shim_is_the_same_entry_points(monkeypatch):
    """tonic_amd/_fastcall (csrc/fastcall.c) binds the per-step tonic_collector_* entries without ctypes:
    same C functions, same arguments, same status codes and the same bytes in the block — and the package
    falls back to the ctypes prototypes with TONIC_AMD_FASTCALL=0."""
    from tonic_amd import _lib
    from tonic_amd.collector import Block
    lib = _lib.load()
    fast = _lib.hot('tonic_collector_synthetic_step')
    assert type(fast).__name__ == 'builtin_function_or_method', 'tonic_amd/_fastcall*.so is not built'
    rng = np.random.RandomState(3)
    rows, actions = rng.standard_normal((7, 5)).astype(np.float32), rng.standard_normal((7, 2)).astype(np.float32)
    one, two = Block(7, 5, 2), Block(7, 5, 2)
    for block, call in ((one, fast), (two, lib.tonic_collector_synthetic_step)):
        block.actions[:] = actions
        assert call(block.address, rows.ctypes.data, None, False) == 0
    for field in ('observations', 'next_observations', 'rewards'):
        assert np.array_equal(getattr(one, field), getattr(two, field)), field
    assert np.array_equal(one.rewards, -(actions * actions).sum(1, dtype=np.float32))
    # explicit actions, a true `ring` with nothing armed, the error path of a bad block
    assert fast(one.address, rows.ctypes.data, actions.ctypes.data, True) == 0
    assert fast(None, rows.ctypes.data, None, False) == lib.tonic_collector_synthetic_step(None, rows.ctypes.data, None, 0) != 0
    for name, args in (('tonic_collector_wait_actions', (None, 0.01)), ('tonic_collector_arm', (None, 0, 0, True)),
                       ('tonic_collector_ppo_step', (None, 0, 0, False)), ('tonic_collector_ring', (one.address,))):
        assert _lib.hot(name)(*args) == getattr(lib, name)(*args), name
    with pytest.raises(TypeError):
        fast(one.address, rows.ctypes.data)
    monkeypatch.setattr(_lib, '_fast', False)
    monkeypatch.setenv('TONIC_AMD_FASTCALL', '0')
    assert _lib.hot('tonic_collector_ring') is lib.tonic_collector_ring
    monkeypatch.setattr(_lib, '_fast', False)


def test_late_rows_reach_the_logger_before_it_reduces_an_epoch():
    """An agent that stores part of an update late (PPO: the critic's rows, whose iterations run
    under the next rollout) registers with `logger.before_dump`: its rows must be stored before the
    epoch is reduced — by this package's logger and by a HOST logger (the reference's, when it owns
    the run), whose module-level `dump` gets the same prologue exactly once."""
    import types
    from tonic_amd.utils import logger

    class Late:
        def __init__(self):
            self.flushed = 0

        def settle(self):
            self.flushed += 1
            logger.store('critic/loss', 1.5)

    late = Late()
    logger.before_dump(late, 'settle')
    try:
        calls = []
        host = types.SimpleNamespace(
            store=lambda *a, **k: calls.append(('store',) + a),
            dump=lambda: calls.append(('dump',)),
            get_current_logger=lambda: host)
        logger.use(host)
        logger.get_current_logger()                  # forwards -> hooks the host's dump
        logger.get_current_logger()                  # ... once
        host.dump()
        assert late.flushed == 1
        assert calls == [('store', 'critic/loss', 1.5), ('dump',)], calls
        host.dump()
        assert late.flushed == 2 and calls[-1] == ('dump',) and len(calls) == 4
    finally:
        logger.use(None)
        logger._before_dump[:] = [(ref, m) for ref, m in logger._before_dump if ref() is not late]
    # an owner that is gone is dropped, not called
    gone = Late()
    logger.before_dump(gone, 'settle')
    count = len(logger._before_dump)
    del gone
    logger._run_before_dump()
    assert len(logger._before_dump) == count - 1


def test_logger_keeps_its_own_copy_of_step_views(tmp_path):
    """`agent.step` / `environment.step` hand out persistent views of the collector block, overwritten in
    place by the next step (deviation 2).  The trainer stores every step's actions for the
    epoch's statistics (trainer.py:71): the logger must reduce what each step held, not N copies of the last
    one — arrays that own their memory are kept as they are, like the reference does."""
    from tonic_amd.utils import logger
    log = logger.Logger(path=str(tmp_path / 'run'))
    block = np.zeros((4, 2), np.float32)
    view = block.view()
    view.setflags(write=False)
    for step in range(3):
        block[:] = step
        log.store('train/action', view, stats=True)
    owned = np.ones(3)
    log.store('other', owned)
    assert log.epoch_dict['other'][0] is owned
    kept = log.epoch_dict['train/action']
    assert [float(k[0, 0]) for k in kept] == [0.0, 1.0, 2.0]
    log._reduce()
    assert log.epoch_dict['train/action/mean'] == 1.0
    assert log.epoch_dict['train/action/min'] == 0.0 and log.epoch_dict['train/action/max'] == 2.0


def test_environments_promise_carry_over_rows(monkeypatch):
    """Sequential / Parallel / SyntheticBatch write `observations` of step t + 1 = `next_observations`
    of step t for every worker that did not reset (distributed.py:41-57) and say so in the block's
    header (tonic_collector_block_carry_over), which lets many-worker steps move each observation
    row over PCIe once; a block made by anybody else promises nothing; TONIC_AMD_CARRY_OVER=0 is the
    developer switch."""
    from tonic_amd import environments
    from tonic_amd.collector import Block
    O, A, W = 5, 2, 6

    def promised(block):
        # int32 header words: magic, version, W (8 bytes), O, A, groups, carry_over
        return int(np.frombuffer(block.memory, np.int32, 8, 0)[7])

    assert promised(Block(W, O, A)) == 0
    env = environments.SyntheticBatch(W, O, A, max_episode_steps=3, termination_probability=0.3)
    env.initialize(seed=0)
    observations = env.start()
    assert promised(env.block) == 1
    for t in range(12):                              # ... and keep the promise
        observations, infos = env.step(np.zeros((W, A), np.float32))
        kept = ~infos['resets']
        assert np.array_equal(observations[kept], infos['observations'][kept])
    sequential = environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=3), 1, W)
    sequential.initialize(seed=0)
    sequential.start()
    assert promised(sequential.block) == 1
    monkeypatch.setenv('TONIC_AMD_CARRY_OVER', '0')
    env = environments.SyntheticBatch(W, O, A)
    env.initialize(seed=0)
    env.start()
    assert promised(env.block) == 0


def test_trainer_bookkeeping(tmp_path):
    import tonic_amd
    from tonic_amd import agents, environments, logger

    class Constant(agents.Agent):
        updates = 0

        def step(self, observations, steps):
            return np.zeros((len(observations), 2), np.float32)

        test_step = step

        def update(self, observations, rewards, resets, terminations, steps):
            self.updates += 1
            assert observations.shape == (3, 4) and resets.dtype == bool

        def save(self, path):
            self.saved = path

    logger.initialize(path=str(tmp_path))
    env = environments.distribute(lambda: environments.Synthetic(4, 2, max_episode_steps=5), 1, 3)
    env.initialize(seed=0)
    test_env = environments.distribute(lambda: environments.Synthetic(4, 2, max_episode_steps=5), 1, 1)
    test_env.initialize(seed=10000)
    agent = Constant()
    trainer = tonic_amd.Trainer(steps=60, epoch_steps=30, save_steps=60, show_progress=False)
    trainer.initialize(agent, env, test_env)
    trainer.run()
    assert agent.updates == 20 and trainer.steps == 60
    assert agent.saved.endswith(os.path.join('checkpoints', 'step_60'))
    rows = open(tmp_path / 'log.csv').read().strip().split('\n')
    header = rows[0].split(',')
    for key in ('train/steps_per_second', 'train/episode_length/mean', 'test/episode_score/mean',
                'train/action/mean', 'train/epochs'):
        assert key in header
    assert len(rows) == 3
    first = dict(zip(header, rows[1].split(',')))
    assert float(first['train/episode_length/mean']) == 5.0       # time-outs every 5 steps
    assert float(first['test/episode_length/mean']) == 5.0


def test_shard_bounds_partition():
    from tonic_amd import parallel
    for total in (256, 10240, 7):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'oracle'))
import numpy_port as port
from tonic_amd import parallel
rank, world = parallel.init_from_env(backend='gloo')
rng = np.random.RandomState(0)
T, W, O, A = 6, 10, 17, 6
params = [rng.normal(size=s).astype(np.float32) * 0.2 for s in
          [(64, O), (64,), (64, 64), (64,), (1, A), (A, 64), (A,)]]
obs = rng.normal(size=(T, W, O)).astype(np.float32)
act = np.clip(rng.normal(size=(T, W, A)), -1, 1).astype(np.float32)
ret = rng.normal(size=(T, W)).astype(np.float32)
val = rng.normal(size=(T, W)).astype(np.float32)
old = (rng.normal(size=(T, W)) - 6).astype(np.float32)
lo, hi = parallel.shard_bounds(W)
raw = (ret - val)[:, lo:hi].astype(np.float64)
mean, std, all_zero = parallel.combine_advantage_moments(
    raw.sum(), (raw * raw).sum(), raw.min(), raw.max(), raw.size)
adv_full = port.normalized_advantages(ret, val)
adv = ((ret - val)[:, lo:hi] - np.float32(mean)) / np.float32(std)
np.testing.assert_allclose(adv, adv_full[:, lo:hi], rtol=1e-5, atol=1e-5)
f = lambda x: x[:, lo:hi].reshape((-1,) + x.shape[2:])
n_local = T * (hi - lo)
grads, stats = port.clipped_ratio_grads(params, f(obs), f(act), adv.reshape(-1), f(old))
sums = torch.tensor(np.concatenate([g.reshape(-1) for g in grads] +
                                   [[stats['loss'], stats['kl'], n_local]]).astype(np.float64))
sums[:-1] *= n_local                       # ranks exchange SUMS, not means
parallel.allreduce_sums(sums)
n_global = float(sums[-1])
assert n_global == T * W
full_g, full_s = port.clipped_ratio_grads(
    params, obs.reshape(-1, O), act.reshape(-1, A), adv_full.reshape(-1), old.reshape(-1))
want = np.concatenate([g.reshape(-1) for g in full_g]).astype(np.float64)
got = sums[:-3].numpy() / n_global
assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-8, np.abs(got - want).max()
assert abs(float(sums[-3]) / n_global - full_s['loss']) < 1e-5
assert abs(float(sums[-2]) / n_global - full_s['kl']) < 1e-5
print('rank', rank, 'ok')
'''


def test_two_rank_gradient_sum_exchange_gloo(tmp_path):
    """world_size 2 over gloo: sharding the worker axis and all-reducing gradient SUMS gives
    the single-process full-batch gradient (the protocol the RCCL path uses)."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', WORLD_SIZE='2',
               OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f'rank {r} ok' in out


CHOICE_WORKER = '''
import sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from tonic_amd import parallel
rank, world = parallel.init_from_env(backend='gloo')
assert world == 2
# every verdict is the AND over the ranks, with the first objecting rank's reason, on every rank
assert parallel._agree(True) == (True, '')
ok, why = parallel._agree(rank != 1, 'no window on this rank')
assert not ok and why == 'rank 1: no window on this rank', (ok, why)
# auto: no GPU here -> every rank declines together, with the reason; the learner then uses the
# process group.  The decision is taken once.
assert parallel.one_shot(1000) is None
choice = parallel.allreduce_choice()
assert choice['kind'] == 'rccl' and 'without a GPU' in choice['reason'], choice
assert parallel.one_shot(5000) is None and parallel.allreduce_choice() is choice
dist.barrier()
print('rank', rank, 'ok')
'''


def test_exchange_choice_is_unanimous_and_recorded_gloo(tmp_path):
    """parallel.one_shot in its automatic mode on two CPU ranks: the ranks agree on every step of the
    decision (`_agree`: the AND over the ranks + the first objecting rank's reason, identical
    everywhere), decline the one-shot all-reduce together where it cannot work (no GPU) and keep the
    reason (`allreduce_choice`) — the logic that selects tonic_allreduce_f32 on a multi-GPU node."""
    script = tmp_path / 'choice.py'
    script.write_text(CHOICE_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', WORLD_SIZE='2',
               OMP_NUM_THREADS='1')
    env['TONIC_AMD_ALLREDUCE'] = 'auto'            # (the default is the process group: see parallel.one_shot)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f'rank {r} ok' in out


CFG4_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import tonic_amd
from tonic_amd import parallel
from tonic_amd.torch.agents import shard_noise
rank, world = parallel.init_from_env(backend='gloo')
# BASELINE config 4 (SURVEY 8 cfg table): Buffer() -> max_size = 1e6 // 512 = 1953 rows of 512 workers, the
# reference's default batch of 100, 50 iterations per update, humanoid-walk actions (A = 21)
W_global, B, iterations, A = 512, 100, 50, 21
buf = tonic_amd.replays.Buffer(size=int(1e6), batch_size=B, batch_iterations=iterations)
rows = int(1e6) // W_global
assert rows == 1953
lo, hi = parallel.shard_bounds(W_global)
buf.num_workers, buf.global_workers, buf.rank, buf.world = hi - lo, W_global, rank, world
buf.max_size, buf.size = rows, rows
# the reference's global index stream (buffers.py:86) and the learner's noise, the same on every rank
stream = np.random.RandomState(0)
indices = np.stack([stream.randint(rows * W_global, size=B) for _ in range(iterations)])
eps = np.random.RandomState(1).standard_normal((iterations, 1, B, A)).astype(np.float32)
local, positions, counts = buf.shard_indices(indices)
mine = shard_noise(eps, positions, counts, B)
# (1) the per-rank parts partition every batch
owned = torch.zeros(iterations, B, dtype=torch.int64)
for it in range(iterations):
    owned[it, positions[it, :counts[it]]] = 1
dist.all_reduce(owned)
assert bool((owned == 1).all()), 'every sample of a global batch belongs to exactly one rank'
totals = torch.tensor(counts)
dist.all_reduce(totals)
assert bool((totals == B).all())
# (2) a local index addresses the same transition in the [1953, 256] shard: transition ids as payload
ids = np.arange(rows * W_global, dtype=np.int64).reshape(rows, W_global)[:, lo:hi].reshape(-1)
for it in range(iterations):
    c = counts[it]
    assert np.array_equal(ids[local[it, :c]], indices[it, positions[it, :c]])
    assert np.array_equal(mine[it, 0, :c], eps[it, 0, positions[it, :c]])
# (3) the exchange of one optimizer step at this configuration's sizes: the critics' 177 666 + 8 gradient
#     sums and the actor's 88 597 + 8 (SURVEY 8e), summed over the ranks in one all-reduce each
for floats in (177666 + 8, 88597 + 8):
    sums = torch.arange(floats, dtype=torch.float32) % 1009.0 * (rank + 1)
    parallel.allreduce_sums(sums)
    want = torch.arange(floats, dtype=torch.float32) % 1009.0 * (world * (world + 1) // 2)
    assert torch.equal(sums, want)
print('rank', rank, 'ok', int(counts.min()), int(counts.max()))
'''


def test_cfg4_shard_path_at_baseline_shapes_gloo(tmp_path):
    """BASELINE config 4's sharded learner on two CPU ranks at its REAL shapes — the global [1953, 512] Buffer
    split into [1953, 256] shards, the reference's default batch of 100, 50 iterations: the global index stream
    splits into per-rank parts that partition every batch and address the right transitions, the noise rows
    follow their samples, and the per-step exchange (177 674 / 88 605 floats) sums over the ranks."""
    script = tmp_path / 'cfg4.py'
    script.write_text(CFG4_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', WORLD_SIZE='2',
               OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f'rank {r} ok' in out


def test_buffer_shard_indices_partition_the_global_batch():
    """SURVEY §8e, off-policy: every rank draws the same GLOBAL index stream and keeps the samples
    whose worker column it owns.  The per-rank parts must partition the batch, and the local
    flat index must address the same transition in the rank's [rows, W_local] shard."""
    import tonic_amd
    rng = np.random.RandomState(0)
    world, w_local, rows, batch, iterations = 4, 3, 50, 64, 5
    w_global = world * w_local
    indices = rng.randint(rows * w_global, size=(iterations, batch))
    global_ids = np.arange(rows * w_global).reshape(rows, w_global)       # a "transition id" buffer
    seen = np.zeros((iterations, batch), int)
    for rank in range(world):
        buf = tonic_amd.replays.Buffer(size=rows * w_global, batch_size=batch)
        buf.num_workers, buf.global_workers, buf.rank, buf.world = w_local, w_global, rank, world
        local, positions, counts = buf.shard_indices(indices)
        shard = global_ids[:, rank * w_local:(rank + 1) * w_local]        # this rank's columns
        for it in range(iterations):
            c = counts[it]
            pos = positions[it, :c]
            assert np.array_equal(shard.reshape(-1)[local[it, :c]], indices[it, pos])
            assert np.all(np.diff(pos) > 0), 'batch order is kept inside a shard'
            seen[it, pos] += 1
    assert np.all(seen == 1), 'every sample of the global batch is owned by exactly one rank'


def test_offpolicy_padded_parameter_layout_keeps_state_dict():
    """Off-policy blocks: padded weight rows (include/tonic_hip.h), parameters are strided views,
    state_dict / checkpoints keep the reference's shapes, the C side agrees on the block lengths."""
    import io
    import torch
    import tonic_amd.torch as tt
    from tonic_amd import _lib
    from tonic_amd.environments import Box
    lib = _lib.load()
    assert [lib.tonic_mlp_weight_stride(c) for c in (256, 111, 119, 88, 96, 64, 1)] == \
        [260, 112, 120, 88, 100, 68, 4]

    def packed():
        model = tt.agents.SAC().model
        model.initialize(Box(-np.inf, np.inf, (111,)), Box(-1, 1, (8,)))
        before = {k: v.clone() for k, v in model.state_dict().items()}
        return model.pack('cpu'), before

    torch.manual_seed(0)
    model, before = packed()
    after = model.state_dict()
    assert all(torch.equal(before[k], after[k]) and before[k].shape == after[k].shape for k in before)
    assert model.actor.torso.model[2].weight.stride() == (260, 1)
    assert model.flat_actor.count == lib.tonic_mlp_actor_param_count(111, 256, 8, 2)
    assert model.flat_critics.count == 2 * lib.tonic_q_critic_param_count(111, 8, 256)
    assert model.flat_online.numel() == model.flat_actor.count + model.flat_critics.count
    live = sum(int((p != 0).sum()) for p in model.online_variables)
    assert int((model.flat_online != 0).sum()) == live            # the padding is zero
    blob = io.BytesIO()
    torch.save(model.state_dict(), blob)
    blob.seek(0)
    other, _ = packed()
    other.load_state_dict(torch.load(blob))
    assert all(torch.equal(other.state_dict()[k], before[k]) for k in before)
    assert int((other.flat_online != 0).sum()) == live
    # deterministic head with an odd action count (TD3 on humanoid-walk shapes), single critic (DDPG)
    for agent_cls, o_dim, a_dim, critics in ((tt.agents.TD3, 67, 21, 2), (tt.agents.DDPG, 17, 6, 1)):
        model = agent_cls().model
        model.initialize(Box(-np.inf, np.inf, (o_dim,)), Box(-1, 1, (a_dim,)))
        model.pack('cpu')
        assert model.flat_actor.count == lib.tonic_mlp_actor_param_count(o_dim, 256, a_dim, 1)
        assert model.flat_critics.count == critics * lib.tonic_q_critic_param_count(o_dim, a_dim, 256)
        assert model.flat_actor.flat.data_ptr() == model.flat_online.data_ptr()
        assert all(p.data_ptr() % 16 == 0 for p in model.online_variables)


def test_ranks_partition_the_single_process_run(monkeypatch):
    """One process per GPU (RANK / WORLD_SIZE of the launcher): the workers of N ranks are seeded
    like the workers of one process, the rows a rank keeps of the global exploration draws
    (TONIC_AMD_GLOBAL_NOISE=1) are its workers' rows of the single-process stream, its own stream
    (default) differs from rank to rank, and the Trainer counts the steps of the whole job."""
    import tonic_amd
    from tonic_amd import environments, explorations

    def observations(rank, world, workers):
        monkeypatch.setenv('RANK', str(rank))
        monkeypatch.setenv('WORLD_SIZE', str(world))
        env = environments.distribute(lambda: environments.Synthetic(5, 2), 1, workers)
        env.initialize(seed=3)
        first = env.start().copy()
        second = env.step(np.zeros((workers, 2), np.float32))[0].copy()
        return np.stack([first, second], 1)
    whole = observations(0, 1, 8)
    assert np.array_equal(np.concatenate([observations(r, 2, 4) for r in range(2)]), whole)
    assert np.array_equal(np.concatenate([observations(r, 4, 2) for r in range(4)]), whole)

    policy = lambda obs: np.zeros((len(obs), 2), np.float32)
    space = environments.Box(-1, 1, (2,))

    def actions(rank, world, workers, global_noise, cls):
        monkeypatch.setenv('RANK', str(rank))
        monkeypatch.setenv('WORLD_SIZE', str(world))
        monkeypatch.setenv('TONIC_AMD_GLOBAL_NOISE', '1' if global_noise else '0')
        noise = cls(start_steps=2)
        noise.initialize(policy, space, seed=9)
        return np.stack([noise(np.zeros((workers, 5), np.float32), steps) for steps in range(6)], 1)
    for cls in (explorations.NoActionNoise, explorations.NormalActionNoise,
                explorations.OrnsteinUhlenbeckActionNoise):
        whole = actions(0, 1, 6, False, cls)
        parts = [actions(r, 3, 2, True, cls) for r in range(3)]
        assert np.array_equal(np.concatenate(parts), whole), cls.__name__
        own = [actions(r, 3, 2, False, cls) for r in range(3)]
        assert not np.array_equal(own[0][:, :3], own[1][:, :3])          # warm-up draws differ

    class Counting(tonic_amd.agents.Agent):
        def __init__(self):
            self.steps = []

        def step(self, observations, steps):
            self.steps.append(steps)
            return np.zeros((len(observations), 2), np.float32)

        def test_step(self, observations, steps):
            return self.step(observations, steps)
    monkeypatch.setenv('RANK', '1')
    monkeypatch.setenv('WORLD_SIZE', '4')
    env = environments.distribute(lambda: environments.Synthetic(5, 2), 1, 3)
    env.initialize(seed=0)
    agent = Counting()
    trainer = tonic_amd.Trainer(steps=48, epoch_steps=24, save_steps=1000, show_progress=False)
    trainer.initialize(agent, env)
    trainer.run()
    assert agent.steps == [0, 12, 24, 36]


def test_shard_noise_keeps_the_owned_rows_in_sample_major_order():
    """agents.shard_noise (several ranks, global noise stream): element by element against the
    definition, for one draw per state and for MPO's S draws per state; the ranks' parts together
    are the global draws."""
    from tonic_amd.torch.agents import shard_noise
    rng = np.random.RandomState(4)
    for S in (1, 5):
        iterations, draws, B, A, world = 3, 2, 7, 2, 3
        eps = rng.standard_normal((iterations, draws, S * B, A)).astype(np.float32)
        owner = rng.randint(world, size=(iterations, B))
        rebuilt = np.zeros_like(eps)
        for rank in range(world):
            counts = (owner == rank).sum(1)
            positions = np.zeros((iterations, B), int)
            for it in range(iterations):
                positions[it, :counts[it]] = np.nonzero(owner[it] == rank)[0]
            local = shard_noise(eps, positions, counts, B)
            assert local.shape == eps.shape
            for it in range(iterations):
                c = counts[it]
                assert not local[it, :, S * c:].any()
                for s in range(S):
                    for j in range(c):
                        m = positions[it, j]
                        assert np.array_equal(local[it, :, s * c + j], eps[it, :, s * B + m])
                        rebuilt[it, :, s * B + m] = local[it, :, s * c + j]
        assert np.array_equal(rebuilt, eps)
