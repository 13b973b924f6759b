"""A2C / PPO / TRPO / DDPG / TD3 / SAC / D4PG / MPO on the HIP engine — API of ``tonic/torch/agents/*.py``.

``step`` / ``update`` keep the reference signatures (NumPy in, NumPy out) so
``tonic.Trainer`` drives the agents unchanged.  The on-policy agents talk to the GPU through
the pinned-host collector (``tonic_amd.collector``, ``tonic_collector_*``): per environment
step ONE fused launch — policy forward + sample + log-prob of the block's observations,
Segment row store, ``MeanStd.record``, and the deferred store of the previous step's outcome —
reads the environment's shared block in place and writes the actions back into it; the host
waits on a completion word, not on the stream, and draws the next step's noise meanwhile.
``update`` only marks the outcome that sits in the block as pending.  Every ``Segment.size``
steps ``_update`` enqueues the whole learner update (2 x value forward, GAE scan,
``batch_iterations`` x [actor grad+Adam, critic grad+Adam]) without any host synchronisation —
the KL early stop of ppo.py:45-46 is a device flag — and reads the logged statistics back once.

The action noise is drawn on the host with ``torch.randn`` from the global CPU generator,
which consumes the same stream as the reference's ``Normal.sample()``, so
runs with equal seeds follow the reference's trajectory up to float32 rounding.  The draw for
step t+1 is made while the GPU works on step t; ``test_step`` (which draws from the same
generator, a2c.py:87-90) first rewinds the generator to where the reference would be.
"""
import ctypes
import os
import random
import time
import weakref

import numpy as np
import torch

from tonic_amd import _lib, agents, explorations, logger, parallel, replays
from tonic_amd.collector import Block, Collector
from tonic_amd.torch import models, normalizers, updaters


def default_model():
    """tonic/torch/agents/a2c.py:7-17."""
    return models.ActorCritic(
        actor=models.Actor(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((64, 64), torch.nn.Tanh),
            head=models.DetachedScaleGaussianPolicyHead()),
        critic=models.Critic(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((64, 64), torch.nn.Tanh),
            head=models.ValueHead()),
        observation_normalizer=normalizers.MeanStd())


_SIDE_STREAMS = {}
_ARMED_GATES = weakref.WeakSet()      # agents whose critic chain waits behind a tonic_stream_gate


class _DevicePointer:
    """A raw device address where the entry-point wrappers expect a contiguous float32 tensor (`_lib.ptr`): fields
    of a page-locked collector block as the GPU addresses them."""

    def __init__(self, address, shape):
        self.address, self.shape = address, tuple(shape)

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.address


def _side_stream(name):
    """One stream per purpose and device for the whole process, not one per agent: HIP maps the
    streams of a process onto a handful of hardware queues in creation order, and two streams on one
    queue run one after the other — an agent whose critic stream landed on its collector's queue
    would get no overlap at all (seen in a process that had built a dozen agents before)."""
    key = (name, torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream()
    return _SIDE_STREAMS[key]


def _device():
    if not torch.cuda.is_available():
        raise _lib.TonicHipError(
            'tonic_amd agents need a ROCm GPU: the learner runs in hand-written HIP kernels '
            'and there is deliberately no CPU fallback')
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    device = torch.device('cuda', local_rank if local_rank < torch.cuda.device_count() else 0)
    parallel.bind_near_gpu(device)             # (the collect loop is PCIe round trips: stay on the GPU's socket)
    return device


class Agent(agents.Agent):
    """tonic/torch/agents/agent.py:10-26 (seeding and .pt checkpoints)."""

    def initialize(self, seed=None):
        parallel.init_from_env()          # one process per GPU under torch.distributed.run; else no-op
        self.seed = seed
        if seed is not None:
            np.random.seed(seed)
            random.seed(seed)
            torch.manual_seed(seed)

    def _replicate(self, buffers, own_noise):
        """Several ranks: rank 0's parameters everywhere (replicas must not hinge on equal seeds),
        then the ACTING noise of this rank's workers gets a stream of its own (seed + rank), unless
        TONIC_AMD_GLOBAL_NOISE asks for the single-process stream (parallel.global_noise):
        `own_noise` — the on-policy agents, whose torch generator feeds nothing but the action
        noise — re-seeds the global generator; the off-policy agents keep the global generator for
        the update's replicated noise stream (`_draw_noise`, which must stay in step across ranks)
        and act from a dedicated per-rank generator — otherwise worker i of every rank would
        explore with the very same draws."""
        self.rank, self.world = parallel.rank(), parallel.world_size()
        self.global_noise = self.world > 1 and parallel.global_noise()
        self._acting_generator = None
        if self.world == 1:
            return
        parallel.broadcast_from_first(buffers)
        if self.global_noise:
            return
        if own_noise:
            if self.seed is not None and self.rank > 0:
                torch.manual_seed(self.seed + self.rank)
            return
        self._acting_generator = torch.Generator()
        if self.seed is not None:
            self._acting_generator.manual_seed(self.seed + self.rank)
        else:
            self._acting_generator.seed()

    def _randn(self, workers, width, out=None):
        """Standard-normal action noise for this rank's `workers` workers (global_noise: rows
        [rank * workers, (rank + 1) * workers) of the draw for all of them)."""
        if not getattr(self, 'global_noise', False):
            generator = getattr(self, '_acting_generator', None)
            if out is not None:
                return torch.randn(workers, width, out=out, generator=generator)
            return torch.randn(workers, width, generator=generator)
        rows = torch.randn(self.world * workers, width)[self.rank * workers:(self.rank + 1) * workers]
        return rows if out is None else out.copy_(rows)

    def save(self, path):
        path = path + '.pt'
        logger.log(f'\nSaving weights to {path}')
        os.makedirs(os.path.dirname(path), exist_ok=True)
        # CPU tensors with the reference's key layout: loadable by the reference and tonic.play.
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, path)

    def load(self, path):
        path = path + '.pt'
        logger.log(f'\nLoading weights from {path}')
        self.model.load_state_dict(torch.load(path, map_location='cpu'))


class _Staging:
    """One [W, ...] record per environment step in PINNED host memory that the kernels read and
    write IN PLACE (page-locked memory is mapped into the GPU's address space at the same
    address): no hipMemcpyAsync in either direction, no stream synchronisation — the host fills
    the fields, enqueues the kernels with these very pointers, records an event behind them and
    spins on it.  (`TONIC_AMD_STAGING=copy` restores explicit H2D / D2H copies through a device
    mirror.)"""

    def __init__(self, fields, device):
        self.offsets, total = {}, 0
        for name, shape in fields:
            size = int(np.prod(shape))
            self.offsets[name] = (total, size, tuple(shape))
            total += (size + 63) // 64 * 64              # fields on 256-byte boundaries
        self.host = torch.zeros(total, dtype=torch.float32).pin_memory()
        self.host_np = self.host.numpy()
        self.mapped = os.environ.get('TONIC_AMD_STAGING', 'mapped') != 'copy'
        if self.mapped:      # kernels must see the block at the address the host uses
            seen = _lib.load().tonic_host_device_pointer(self.host.data_ptr())
            self.mapped = seen == self.host.data_ptr()
        self.device = self.host if self.mapped else torch.empty(total, dtype=torch.float32,
                                                                device=device)
        self.done = None             # event behind the last kernels / copies that touched `host`

    def host_view(self, name):
        start, size, shape = self.offsets[name]
        return self.host_np[start:start + size].reshape(shape)

    def device_view(self, name):
        """What the kernels get: the pinned field itself, or its device mirror."""
        start, size, shape = self.offsets[name]
        return self.device[start:start + size].view(shape)

    def writable(self):
        """Blocks until the work recorded by mark() has consumed the pinned block: the host may
        not overwrite it earlier (nothing else orders consecutive update() calls while an agent
        is still warming up and never reads anything back)."""
        if self.done is not None:
            self.done.synchronize()
        return self

    def upload(self):
        if not self.mapped:
            self.device.copy_(self.host, non_blocking=True)

    def download(self):
        if not self.mapped:
            self.host.copy_(self.device, non_blocking=True)

    def mark(self):
        """Records the event behind everything enqueued so far that reads / writes this block."""
        if self.done is None:
            self.done = torch.cuda.Event()
        self.done.record()

    def wait(self):
        """Spins until the marked work is complete (results written in place are visible)."""
        done = self.done
        while not done.query():
            pass


class _NoiseAhead:
    """The action noise of the collect loop, off its critical path (a2c.py:81 draws
    ``Normal.sample()`` = one ``torch.randn(W, A)`` per environment step: 5 - 9 us of host time per
    step at W = 256).  The draws of DEPTH steps are made by ONE call — the generator's stream is the
    same: ``normal_`` fills its output with consecutive uniforms and transforms them in blocks of
    16, so a [DEPTH * W, A] draw equals DEPTH consecutive [W, A] draws bit for bit as long as W * A
    is a multiple of 16 — into one of two buffers; a helper thread fills the next buffer (the GIL
    is released inside the draw) while the loop consumes the current one, 0.7 us per step for the
    copy into the block's noise slot.  The generator runs up to 2 * DEPTH steps ahead of the noise
    the agent has consumed; ``rewind`` puts it back where the reference's stream is (state at the
    start of the current buffer, re-advanced by the rows consumed) before anybody else — a test
    episode — draws from it.  Shapes the block draw does not reproduce (W * A not a multiple of
    16), several ranks slicing a global draw, or TONIC_AMD_NOISE_AHEAD=0: one draw per step with
    the generator state kept before it, as before."""

    DEPTH = 64

    def __init__(self, agent, workers, width):
        import threading
        import weakref
        # (a weak reference: the helper thread keeps THIS object alive, and must not keep the
        #  agent — its Segment, collector and page-locked block — alive with it)
        self._agent, self.workers, self.width = weakref.ref(agent), workers, width
        n = workers * width
        self.bulk = (n % 16 == 0 and n >= 16 and not agent.global_noise
                     and os.environ.get('TONIC_AMD_NOISE_AHEAD', '1') != '0')
        self.generator = getattr(agent, '_acting_generator', None)
        self.mark = None                    # per-step mode: generator state before the last draw
        if not self.bulk:
            return
        self.buffers = [torch.empty(self.DEPTH * workers, width) for _ in range(2)]
        self.rows = [b.numpy().reshape(self.DEPTH, workers, width) for b in self.buffers]
        self.states = [None, None]          # generator state before each buffer was drawn
        self.current, self.taken, self.valid, self.ahead = 0, self.DEPTH, False, False
        self.filling = False                # the helper owns buffers[current ^ 1]
        self.pending = False                # a fill has been asked for and not started yet
        self.request, self.ready = threading.Event(), threading.Event()
        self.closed = False
        self.thread = threading.Thread(target=self._helper, daemon=True, name='tonic-noise-ahead')
        self.thread.start()
        weakref.finalize(agent, self.close)
        # (a daemon thread that is inside torch.randn when the interpreter finalises is killed in
        #  C++ code — "terminate called without an active exception": end the helpers first)
        _NoiseAhead._live.add(self)
        if not _NoiseAhead._at_exit:
            import atexit
            atexit.register(_NoiseAhead._end_all)
            _NoiseAhead._at_exit = True

    _live, _at_exit = weakref.WeakSet(), False

    @staticmethod
    def _end_all():
        for helper in list(_NoiseAhead._live):
            helper.close()
            helper.thread.join(timeout=5.0)

    def close(self):
        """Ends the helper thread (the agent was closed or collected)."""
        if self.bulk and not self.closed:
            self.closed = True
            self.request.set()

    def _state(self):
        return self.generator.get_state() if self.generator is not None else torch.get_rng_state()

    def _set_state(self, state):
        if self.generator is not None:
            self.generator.set_state(state)
        else:
            torch.set_rng_state(state)

    def _fill(self, index):
        self.states[index] = self._state()
        torch.randn(self.buffers[index].shape, out=self.buffers[index], generator=self.generator)

    def _helper(self):
        while True:
            self.request.wait()
            self.request.clear()
            if self.pending:                # (a fill asked for before close() is still delivered)
                self.pending = False
                self._fill(self.current ^ 1)
                self.ready.set()
            if self.closed:
                return

    def _settle(self):
        if self.filling:
            self.ready.wait()
            self.ready.clear()
            self.filling = False

    def take(self, out):
        """The next step's draws -> `out` (the block's noise slot, a NumPy view [W, A])."""
        if not self.bulk:
            self.mark = self._state()
            self._agent()._randn(self.workers, self.width, out=torch.from_numpy(out))
            return
        if self.taken == self.DEPTH:
            if self.valid and self.ahead:   # the buffer drawn ahead takes over
                self._settle()
                self.current ^= 1
            else:                           # first use / after a rewind / no helper: draw here
                self._fill(self.current)
                self.valid = True
            self.taken = 0
            self.ahead = not self.closed    # (a closed helper draws nothing ahead)
            if self.ahead:
                self.filling = True         # the helper draws the buffer after this one
                self.pending = True
                self.request.set()
        np.copyto(out, self.rows[self.current][self.taken])
        self.taken += 1

    def rewind(self, unconsumed):
        """Generator back to the state right after the draws of the steps the agent has executed
        (`unconsumed`: 1 if the last `take` was for a step that has not run yet)."""
        if not self.bulk:
            if unconsumed and self.mark is not None:
                self._set_state(self.mark)
            return
        if not self.valid:
            return
        self._settle()
        consumed = self.taken - unconsumed
        self._set_state(self.states[self.current])
        if consumed > 0:
            torch.randn(consumed * self.workers, self.width, generator=self.generator)
        self.taken, self.valid, self.ahead = self.DEPTH, False, False


class A2C(Agent):
    """Acting / storing half shared by the on-policy agents (a2c.py:20-99)."""

    def __init__(self, model=None, replay=None, actor_updater=None, critic_updater=None):
        self.model = model or default_model()
        self.replay = replay or replays.Segment()
        self.actor_updater = actor_updater or updaters.StochasticPolicyGradient()
        self.critic_updater = critic_updater or updaters.VRegression()

    def initialize(self, observation_space, action_space, seed=None):
        super().initialize(seed=seed)
        self.device = _device()
        if not getattr(self, '_holds_affinity', False):
            self._holds_affinity = True
            parallel.hold_affinity()
        self.lib = _lib.load()
        self.model.initialize(observation_space, action_space)       # CPU init: seed parity
        self.model.pack(self.device)
        if self.model.observation_normalizer:
            self.model.observation_normalizer.attach(self.device)
        self.replay.initialize(seed, device=self.device)
        self.actor_updater.initialize(self.model)
        self.critic_updater.initialize(self.model)
        self.observation_size = observation_space.shape[0]
        self.action_size = action_space.shape[0]
        self._replicate([self.model.flat_actor.flat, self.model.flat_critic.flat], own_noise=True)
        self._collector = self._block = None
        self._speculated = self._eps_ahead = self._block_fed = self._armed = False
        self._critic_pending = self._rollout_behind = None       # (PPO: see _update)
        self._host_rollout = False
        # 3 (default): the act kernel stays RESIDENT for a rollout and the host pushes each step's
        # command, observation and noise rows into a device window (the collector falls back to 2 where
        # it cannot or should not: see tonic_collector_create); 2: resident, the kernel pulls everything
        # from the page-locked block over PCIe; 0: the same kernel launched once per step;
        # 1: hipMemcpyAsync around it (profiles/r02_collector_latency.md: 10.4 / 13.9 / 28.4 us per
        # round trip for 2 / 0 / 1)
        self.transport = int(os.environ.get('TONIC_AMD_COLLECTOR_TRANSPORT', '3'))

    # ------------------------------------------------------------------ acting
    def _io(self, workers):
        W, O, A = workers, self.observation_size, self.action_size
        return (_Staging([('observations', (W, O)), ('eps', (W, A))], self.device),
                _Staging([('actions', (W, A)), ('log_probs', (W,))], self.device))

    def _act(self, observations, stage_in, stage_out, want_log_probs):
        """Stand-alone forward + sample (test episodes): staging copies and a stream sync."""
        W, A = observations.shape[0], self.action_size
        stage_in.writable()
        stage_out.writable()
        stage_in.host_view('observations')[:] = observations
        # Same generator draw as Normal.sample() in the reference (a2c.py:81).
        # (test episodes run on every rank alike: plain draws; training workers are a shard)
        draw = self._randn if want_log_probs else torch.randn
        stage_in.host_view('eps')[:] = draw(W, A).numpy()
        stage_in.upload()
        if getattr(self.actor_updater, 'stock', False):
            # any MLP(sizes, activation): the policy's forward as stock torch operators on the
            # device, the sample and its log-probability as a2c.py:75-85 forms them
            # (mapped staging hands the kernels pinned host fields: torch operators want device
            #  tensors, and write their results back with an asynchronous copy)
            with torch.no_grad():
                observations = stage_in.device_view('observations').to(self.device, non_blocking=True)
                eps = stage_in.device_view('eps').to(self.device, non_blocking=True)
                distribution = self.model.actor(observations)
                actions = distribution.loc + distribution.scale * eps
                stage_out.device_view('actions').copy_(actions, non_blocking=True)
                if want_log_probs:
                    stage_out.device_view('log_probs').copy_(
                        distribution.log_prob(actions).sum(dim=-1), non_blocking=True)
            stage_out.download()
            stage_out.mark()
            stage_in.done = stage_out.done
            stage_out.wait()
            return stage_out.host_view('actions').copy()
        p = _lib.ptr
        torso = getattr(self.actor_updater, 'torso', None)
        if torso is not None:
            need = self.lib.tonic_ppo_torso_workspace_bytes(W, self.observation_size, A, 1, torso[0], torso[1])
        else:
            need = self.lib.tonic_ppo_workspace_bytes(W, self.observation_size, A, 1)
        if getattr(self, '_act_workspace', None) is None or self._act_workspace.numel() < need:
            self._act_workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        if torso is not None:
            _lib.check(self.lib.tonic_ppo_act_torso(
                *torso, p(self.model.flat_actor.flat), p(stage_in.device_view('observations')),
                p(stage_in.device_view('eps')), p(stage_out.device_view('actions')),
                p(stage_out.device_view('log_probs')) if want_log_probs else None,
                W, self.observation_size, A, p(self._act_workspace), self._act_workspace.numel(),
                _lib.current_stream()), 'tonic_ppo_act_torso')
        else:
            _lib.check(self.lib.tonic_ppo_act_wide(
                p(self.model.flat_actor.flat), p(stage_in.device_view('observations')),
                p(stage_in.device_view('eps')), p(stage_out.device_view('actions')),
                p(stage_out.device_view('log_probs')) if want_log_probs else None,
                W, self.observation_size, A, p(self._act_workspace), self._act_workspace.numel(),
                _lib.current_stream()), 'tonic_ppo_act_wide')
        stage_out.download()
        stage_out.mark()
        stage_in.done = stage_out.done
        stage_out.wait()
        return stage_out.host_view('actions').copy()

    def _bind(self, observations):
        """First step (or a new worker count): find the environment's block — the arrays of
        tonic_amd.environments ARE views of one — or make a private one, page-lock it and point
        the collector at the Segment."""
        W, O, A = observations.shape[0], self.observation_size, self.action_size
        block = Block.owner_of(observations) or Block(W, O, A)
        self._block = block
        self._collector = Collector.for_block(block, self.transport)
        self.transport_in_effect = self._collector.transport
        replay = self.replay
        if replay.buffers is None or replay.num_workers != W:
            replay._allocate(W, O, A)
        self._eps = (torch.from_numpy(block.eps[0]), torch.from_numpy(block.eps[1]))
        noise = getattr(self, '_noise', None)
        if noise is None or (noise.workers, noise.width) != (W, A):
            if noise is not None:
                noise.rewind(0)
            self._noise = _NoiseAhead(self, W, A)
        self._slot, self._eps_ahead, self._pending, self._rollout_open = 0, False, False, False
        # update() may issue the NEXT step's launch itself (see there): whether it did, and whether
        # the caller has been handing over the block's own arrays so far
        self._speculated, self._block_fed = False, False
        self._speculate = os.environ.get('TONIC_AMD_SPECULATE', '1') != '0'
        # step() may leave the next command with the environment (Block.ring issues it)
        self._armed = False
        self.steps_issued_by_environment = 0
        self._arm_steps = (self._speculate and not self.model.return_normalizer
                           and os.environ.get('TONIC_AMD_ARM', '1') != '0')

    # -- shapes beyond the fused act kernel (O > 32 or A > 8): staged copies + separate launches
    def _wide(self):
        """Shapes beyond the fused act kernel go through the collector as well (layer-by-layer
        launches per step on the mapped block, csrc/mlpwide.hip wide_collect_step);
        TONIC_AMD_WIDE_STAGED=1 keeps the older path of staged copies and separate launches."""
        # (torsos outside the kernels' shapes act through stock torch operators, staged as well)
        # (... and torsos on the layer-by-layer HIP path, tonic_ppo_act_torso, staged too: the collector's
        #  per-step kernels hold the default torso)
        return getattr(self.actor_updater, 'stock', False) or \
            getattr(self.actor_updater, 'torso', None) is not None or (
                (self.observation_size > 32 or self.action_size > 8)
                and os.environ.get('TONIC_AMD_WIDE_STAGED', '0') == '1')

    def _step_staged(self, observations):
        observations = np.asarray(observations, np.float32)
        W, O = observations.shape[0], self.observation_size
        if getattr(self, '_staged_workers', None) != W:
            self._staged_workers = W
            self._in, self._out = self._io(W)
            self._outcome = _Staging([('next_observations', (W, O)), ('rewards', (W,)),
                                      ('resets', (W,)), ('terminations', (W,))], self.device)
        actions = self._act(observations, self._in, self._out, True)
        self.last_observations, self.last_actions = observations, actions
        return actions

    def _update_staged(self, observations, rewards, resets, terminations):
        stage = self._outcome.writable()
        stage.host_view('next_observations')[:] = observations
        stage.host_view('rewards')[:] = rewards
        stage.host_view('resets')[:] = resets                 # bool -> float32 (segments.py:33)
        stage.host_view('terminations')[:] = terminations
        stage.upload()
        self.replay.store(
            normalizer=self.model.observation_normalizer,
            observations=self._in.device_view('observations'),
            actions=self._out.device_view('actions'),
            next_observations=stage.device_view('next_observations'),
            rewards=stage.device_view('rewards'), resets=stage.device_view('resets'),
            terminations=stage.device_view('terminations'),
            log_probs=self._out.device_view('log_probs'))
        stage.mark()                  # the store kernel reads the pinned fields in place
        self._in.done = self._out.done = stage.done
        if self.replay.ready():
            self._update()

    def step(self, observations, steps):
        # The steady state of a block-fed loop first (every attribute lookup of this call is host
        # time between the simulator's step and the next command to the GPU): update() has issued
        # this step with the block's own observations and the noise drawn ahead.
        if self._speculated and self._eps_ahead and observations is self._block.out_observations:
            block, slot = self._block, self._slot
            self._speculated = False
            self._noise.take(block.eps[slot ^ 1])      # the step after this one
            self._slot = slot ^ 1
            # ... whose command is left with the ENVIRONMENT: it issues it the moment its step
            # record is complete (Block.ring), update() only confirms (see there)
            replay = self.replay
            if self._arm_steps and replay.index + 2 < replay.max_size:
                self._armed = self._collector.arm(replay.index + 1, slot ^ 1, True)
            self._collector.wait_actions()
            self._block_fed = True
            self.last_observations = observations
            self.last_actions = actions = block.out_actions
            return actions
        if self._wide():
            return self._step_staged(observations)
        self._claim()
        block = getattr(self, '_block', None)
        if self._collector is None or block.workers != len(observations):
            self._settle()
            self._bind(observations)
            block = self._block
        fed = observations is block.out_observations or observations is block.observations
        if not fed and self.replay.index == 0 and not self._rollout_open:
            # between two rollouts: another environment of tonic_amd.environments took over (its
            # observations are the view of ANOTHER live block) -> adopt that block, stay zero-copy
            owner = Block.owner_of(observations)
            if owner is not None and owner is not block:
                self._settle()
                eps_ahead = self._eps_ahead and self._eps[self._slot].clone()
                self._bind(observations)
                block, fed = self._block, True
                if eps_ahead is not False:        # the noise drawn ahead moves along (same stream)
                    self._eps[self._slot].copy_(eps_ahead)
                    self._eps_ahead = True
        if not fed:
            np.copyto(block.observations, observations)
        collector = self._collector
        launched = False
        if self._speculated:
            # update() already issued this step with the block's contents and the noise drawn
            # ahead.  Still what the reference would compute?  (Other observations, or a test
            # episode in between whose draws come first in the generator's order: no.)
            self._speculated = False
            if fed and self._eps_ahead:
                launched = True
            else:
                collector.wait_actions()          # discard; a step is idempotent (collector.hip)
        if not launched:
            if not self._rollout_open:
                norm = self.model.observation_normalizer
                behind = None
                if getattr(self, '_rollout_behind', None) is not None:
                    # PPO._update left the critic's iterations running: this rollout writes its
                    # observations to the spare buffer and is ordered behind what it depends on
                    # (the actor, the normaliser) — the current stream waits for the critic as well
                    behind, self._rollout_behind = self._rollout_behind.cuda_stream, None
                    buffers = self.replay.buffers
                    spare = getattr(self, '_spare_observations', None)
                    if spare is None or spare.shape != buffers['observations'].shape:
                        spare = torch.empty_like(buffers['observations'])
                    self._spare_observations, buffers['observations'] = buffers['observations'], spare
                collector.bind_segment(self.replay.buffers,
                                       norm.device_sums if norm is not None else None,
                                       self.replay.max_size)
                collector.begin_rollout(self.model.flat_actor.flat, behind)
                self._rollout_open = True
            if not self._eps_ahead:
                self._noise.take(block.eps[self._slot])                                # a2c.py:81
            collector.ppo_step(self.replay.index, self._slot, self._pending)
            self._pending = False
        slot = self._slot
        # The next step's noise goes into the other slot while the GPU works on this one (drawn
        # ahead by _NoiseAhead; test_step rewinds the generator: its own draws come first in the
        # reference's stream order).
        self._noise.take(block.eps[slot ^ 1])
        self._slot, self._eps_ahead = slot ^ 1, True
        collector.wait_actions()
        # An environment that lives in the block consumes the actions at once (and may take them
        # from the block itself): it gets the block's read-only view; anybody else a fresh array.
        actions = block.out_actions if fed else block.actions.copy()
        self._block_fed = fed
        self.last_observations = observations
        self.last_actions = actions
        return actions

    def _settle(self):
        """Waits out a step that update() issued early and nobody asked for yet (the Segment row it
        wrote lies beyond the rows stored so far and is rewritten by the real step)."""
        self._claim()
        if getattr(self, '_speculated', False):
            self._collector.wait_actions()
            self._speculated = False

    def _claim(self):
        """Takes back a command that step() left with the environment: issued meanwhile = a step
        in flight like one update() issued early; not issued = withdrawn."""
        if getattr(self, '_armed', False):
            self._armed = False
            if self._collector.claim():
                self._speculated = True

    def close(self):
        """Releases what a live agent holds beyond its tensors: waits out a step issued ahead,
        puts the generator back where the reference's stream is, ends the noise helper thread and
        destroys the collector (its stream, its resident kernel, the page-lock on the
        environment's block).  The agent can be initialised again afterwards."""
        if getattr(self, '_collector', None) is not None:
            self._settle()
            noise = getattr(self, '_noise', None)
            if noise is not None:
                noise.rewind(1 if self._eps_ahead else 0)
                noise.close()
                self._noise = None
            self._collector.close()
            self._block._collectors.pop(self._collector.requested, None)
            self._collector = self._block = None
            self._speculated = self._eps_ahead = self._block_fed = False
            self._rollout_behind = None
        if getattr(self, '_holds_affinity', False):
            self._holds_affinity = False
            parallel.release_affinity()            # (bind_near_gpu: the last agent to close restores the CPU mask)

    def test_step(self, observations, steps):
        self._open_gate()       # (the current stream waits for the critic's iterations: let them start)
        noise = getattr(self, '_noise', None)
        if noise is not None:
            self._settle()          # (a step issued ahead reads the slot: let it finish first)
            noise.rewind(1 if self._eps_ahead else 0)
            self._eps_ahead = False
        observations = np.asarray(observations, np.float32)
        if getattr(self, '_test_workers', None) != observations.shape[0]:
            self._test_workers = observations.shape[0]
            self._test_in, self._test_out = self._io(observations.shape[0])
        return self._act(observations, self._test_in, self._test_out, False)

    # ---------------------------------------------------------------- learning
    def update(self, observations, rewards, resets, terminations, steps):
        """a2c.py:58-73.  The outcome stays in the block; the NEXT step's launch (or
        end_rollout) moves it into the Segment row of the step it belongs to."""
        block, replay = self._block, self.replay
        if self._block_fed and observations is block.out_next_observations \
                and rewards is block.out_rewards and resets is block.out_resets \
                and terminations is block.out_terminations and self._eps_ahead and self._speculate \
                and replay.index + 2 < replay.max_size and not self.model.return_normalizer:
            # the steady state of a block-fed loop (the general form is below): the next step's
            # command goes out first, the bookkeeping runs while the GPU works
            index = replay.index
            if self._armed:
                # step() left this command with the environment, which normally has issued it
                # from inside its own step call, before the trainer's loop even got here
                self._armed = False
                if self._collector.claim():
                    self.steps_issued_by_environment += 1
                else:
                    self._collector.ppo_step(index + 1, self._slot, True)
            else:
                self._collector.ppo_step(index + 1, self._slot, True)
            self._speculated = True
            replay.index = index + 1
            gate = self._gate_row
            if gate is not None and index >= gate:
                self._open_gate()
            normalizer = self.model.observation_normalizer
            if normalizer:
                normalizer.new_count += block.workers             # (note_device_rows)
            self._pending = False
            return
        if self._wide():
            return self._update_staged(observations, rewards, resets, terminations)
        self._claim()
        # (identity: the arrays tonic_amd.environments hand out ARE the block's fields)
        if (self._block_fed and observations is block.out_next_observations
                and rewards is block.out_rewards and resets is block.out_resets
                and terminations is block.out_terminations):
            # The environment lives in the block: the outcome is in place and so are the next
            # step's observations (distributed.py:136-155 returns them with these infos).  With
            # its noise drawn ahead, the next step's launch goes out FIRST — the rest of this call,
            # the trainer's bookkeeping and the head of the next agent.step run while the GPU
            # works.  step() issues it again if it turns out to be obsolete.
            if (self._eps_ahead and self._speculate and replay.index + 2 <= replay.max_size
                    and not self._speculated):
                self._collector.ppo_step(replay.index + 1, self._slot, True)
                self._speculated = True
        else:
            self._settle()      # (a step the environment issued reads the block: not under it)
            if (observations is not block.out_next_observations
                    and observations is not block.next_observations):
                np.copyto(block.next_observations, observations)
            if rewards is not block.out_rewards and rewards is not block.rewards:
                np.copyto(block.rewards, rewards)
            if resets is not block.out_resets and resets is not block.resets_bool:
                np.copyto(block.resets, resets)             # bool -> float32 (segments.py:33)
            if terminations is not block.out_terminations and \
                    terminations is not block.terminations_bool:
                np.copyto(block.terminations, terminations)
        if self.model.return_normalizer:
            raise NotImplementedError('return normalisers are not supported (never enabled by '
                                      'the reference defaults)')
        replay.index += 1
        if self._gate_row is not None and replay.index > self._gate_row:
            self._open_gate()
        if self.model.observation_normalizer:
            self.model.observation_normalizer.note_device_rows(block.workers)
        self._pending = not self._speculated
        if replay.ready():
            started = getattr(self, '_rollout_started', None)
            if started is not None:          # host time per environment step of the rollout that ends here
                self._rollout_step_us = (time.perf_counter() - started) * 1e6 / replay.max_size
            self._collector.end_rollout(replay.index - 1)
            self._pending, self._rollout_open = False, False
            # (this rollout came through the collector, which binds the Segment's buffers anew at
            #  every rollout: the update may swap them — see PPO._update)
            self._host_rollout = True
            try:
                self._update()
            finally:
                self._host_rollout = False

    # (PPO: the row of the running rollout at which the gate in front of the critic's iterations opens)
    _gate_row = None

    def _open_gate(self):
        self._gate_row = None

    def _evaluate(self):
        """a2c.py:92-99 on the HBM-resident segment: fills values / next_values in place."""
        b = self.replay.buffers
        critic = self.critic_updater
        critic.forward_values(replays.flatten_batch(b['observations']), b['values'].view(-1))
        critic.forward_values(replays.flatten_batch(b['next_observations']),
                              b['next_values'].view(-1))
        return b['values'], b['next_values']


    def enqueue_update(self):
        """a2c.py:101-127 without a host sync: evaluate, lambda-returns, ONE actor step on the
        full batch, then the critic over `replay.get` (full batch `batch_iterations` times, or
        minibatches).  Returns the statistic rows: [0, 0] the actor's, [1, :] the critic's."""
        replay, actor, critic = self.replay, self.actor_updater, self.critic_updater
        values, next_values = self._evaluate()
        replay.compute_returns(values, next_values)
        updates = replay.updates_per_get()
        if getattr(self, '_infos', None) is None or self._infos.shape[1] != updates:
            self._infos = torch.zeros(2, updates, updaters.INFO_WIDTH, device=self.device)
        self._infos.zero_()
        actor.reset_stop()
        full = tuple(replays.flatten_batch(replay.buffers[k]) for k in replays.segments.LEARNER_KEYS)
        obs, act, raw_adv, log_probs, _ = full
        actor.enqueue(obs, act, raw_adv, replay.adv_stats, log_probs, self._infos[0, 0])
        for it, (obs, _, _, _, returns) in enumerate(replay.learner_batches()):
            critic.enqueue(obs, returns, self._infos[1, it])
        return self._infos

    def _update(self):
        infos = self.enqueue_update().cpu().numpy()          # the only sync of the update
        parallel.check_one_shot()
        for i, key in enumerate(updaters.ACTOR_INFO):
            if key in ('loss', 'kl', 'entropy', 'std'):
                logger.store('actor/' + key, infos[0, 0, i])
        for row in infos[1]:
            logger.store('critic/loss', row[0])
            logger.store('critic/v', row[1])      # mean of the value batch (log-equivalent)
        self.last_infos = infos
        if self.model.observation_normalizer:
            self.model.observation_normalizer.update()


class TRPO(A2C):
    """tonic/torch/agents/trpo.py:7-97.  Acting, storing, evaluation, lambda-returns and the
    critic regression are the A2C / PPO machinery (fused collector, HBM-resident Segment, HIP
    kernels); the actor step is TrustRegionPolicyGradient (stock-torch autograd on the device,
    SURVEY.md §8(f4)).  The behaviour policy's locs / scales are not stored per step
    (trpo.py:27-28,41-43): the parameters are those of the rollout until this update moves them,
    so the updater recomputes them from the stored observations."""

    def __init__(self, model=None, replay=None, actor_updater=None, critic_updater=None):
        super().__init__(model=model, replay=replay,
                         actor_updater=actor_updater or updaters.TrustRegionPolicyGradient(),
                         critic_updater=critic_updater)

    def _update(self):
        replay, critic = self.replay, self.critic_updater
        values, next_values = self._evaluate()
        replay.compute_returns(values, next_values)
        batch = replay.get_full('observations', 'actions', 'log_probs', 'advantages')
        for key, value in self.actor_updater(**batch).items():
            logger.store('actor/' + key, value.numpy())
        updates = replay.updates_per_get()
        infos = torch.zeros(updates, updaters.INFO_WIDTH, device=self.device)
        for it, (obs, _, _, _, returns) in enumerate(replay.learner_batches()):
            critic.enqueue(obs, returns, infos[it])
        infos = infos.cpu().numpy()
        parallel.check_one_shot()
        for row in infos:
            logger.store('critic/loss', row[0])
            logger.store('critic/v', row[1])
        logger.store('critic/iterations', updates)
        self.last_infos = infos
        if self.model.observation_normalizer:
            self.model.observation_normalizer.update()


class PPO(A2C):
    """tonic/torch/agents/ppo.py:7-67."""

    def __init__(self, model=None, replay=None, actor_updater=None, critic_updater=None):
        super().__init__(model=model, replay=replay,
                         actor_updater=actor_updater or updaters.ClippedRatio(),
                         critic_updater=critic_updater)

    def enqueue_update(self):
        """Enqueues one whole learner update on the current stream (no host sync)."""
        self.settle()
        replay, actor, critic = self.replay, self.actor_updater, self.critic_updater
        critic.max_workgroups = self._critic_width()
        values, next_values = self._evaluate()
        replay.compute_returns(values, next_values)
        updates = replay.updates_per_get()
        if getattr(self, '_infos', None) is None or self._infos.shape[1] != updates:
            self._infos = torch.zeros(2, updates, updaters.INFO_WIDTH, device=self.device)
        self._infos.zero_()
        actor.reset_stop()
        world = actor.world_size
        # Full batch (ppo.py:40-47 over segments.py:55-57) or shuffled minibatches
        # (segments.py:58-65); the advantages stay raw and are normalised in-register with the
        # GLOBAL statistics, exactly what get_full computes before the reference slices.
        if not parallel.exchanging():
            for it, (obs, act, raw_adv, log_probs, returns) in enumerate(replay.learner_batches()):
                actor.enqueue_grad(obs, act, raw_adv, replay.adv_stats, log_probs)
                critic.enqueue_grad(obs, returns)
                updaters.enqueue_step_pair(actor, critic, obs.shape[0], replay.adv_stats,
                                           self._infos[0, it], self._infos[1, it])
            return self._infos
        # Several ranks: every iteration exchanges [gradient sums | 8 statistics] of both networks
        # (RCCL over xGMI).  After the KL stop the actor half is zero-filled on every rank alike and
        # its step is skipped by the same device flag everywhere.  Two schedules
        # (TONIC_AMD_EXCHANGE), A/B-measured through RCCL with a one-rank group:
        #   joint (default)  one all-reduce of both halves per iteration, both Adam steps in one
        #                    launch: +8.6 ms per 80-iteration update over the exchange-free path;
        #   overlap          one asynchronous all-reduce per network, each hidden behind the OTHER
        #                    network's fused grad kernel
        #                      actor grad i | AR(actor i) over critic grad i | Adam(actor i)
        #                      critic grad i | AR(critic i) over Adam(actor i) + actor grad i+1 | Adam(critic i)
        #                    : +12.4 ms — the stream hand-overs of 160 collectives cost more than
        #                    the latency of 80 they hide when there is no link latency to hide.
        one_shot = parallel.one_shot(max(actor.count, critic.count) + updaters.INFO_WIDTH)
        if one_shot is not None:
            # tonic_allreduce_f32: one ~10 us launch per exchange, nothing to hide it behind
            for it, (obs, act, raw_adv, log_probs, returns) in enumerate(replay.learner_batches()):
                actor.enqueue_grad(obs, act, raw_adv, replay.adv_stats, log_probs)
                one_shot.all_reduce(actor.grad_sums)
                critic.enqueue_grad(obs, returns)
                one_shot.all_reduce(critic.grad_sums)
                updaters.enqueue_step_pair(actor, critic, obs.shape[0], replay.adv_stats,
                                           self._infos[0, it], self._infos[1, it])
            return self._infos
        all_reduce = torch.distributed.all_reduce
        if os.environ.get('TONIC_AMD_EXCHANGE', 'joint') == 'joint':
            # ONE synchronous all-reduce of [actor sums | 8 | critic sums | 8] per iteration, both
            # optimizer steps in one launch (half the collectives, none of them hidden)
            if getattr(self, '_joint_grads', None) is None:
                na, nc = actor.count + updaters.INFO_WIDTH, critic.count + updaters.INFO_WIDTH
                self._joint_grads = torch.zeros(na + nc, device=self.device)
                actor.share_gradient_buffer(self._joint_grads[:na])
                critic.share_gradient_buffer(self._joint_grads[na:])
            for it, (obs, act, raw_adv, log_probs, returns) in enumerate(replay.learner_batches()):
                actor.enqueue_grad(obs, act, raw_adv, replay.adv_stats, log_probs)
                critic.enqueue_grad(obs, returns)
                all_reduce(self._joint_grads)
                updaters.enqueue_step_pair(actor, critic, obs.shape[0], replay.adv_stats,
                                           self._infos[0, it], self._infos[1, it])
            return self._infos
        pending = None                           # (work handle, rows, info row) of the critic
        for it, (obs, act, raw_adv, log_probs, returns) in enumerate(replay.learner_batches()):
            n = obs.shape[0]
            actor.enqueue_grad(obs, act, raw_adv, replay.adv_stats, log_probs)
            actor_sums = all_reduce(actor.grad_sums, async_op=True)
            if pending is not None:
                pending[0].wait()
                critic.enqueue_step(pending[1], pending[2], allreduce=False)
            critic.enqueue_grad(obs, returns)
            pending = (all_reduce(critic.grad_sums, async_op=True), n, self._infos[1, it])
            actor_sums.wait()
            actor.enqueue_step(n, replay.adv_stats, self._infos[0, it], allreduce=False)
        if pending is not None:
            pending[0].wait()
            critic.enqueue_step(pending[1], pending[2], allreduce=False)
        return self._infos

    # -- the critic's iterations under the next rollout ---------------------------------------
    # The two networks of an update share nothing (ppo.py:33-46: advantages and returns are formed
    # before the first iteration), and a rollout needs the ACTOR only — the critic is not asked
    # again before the next update's evaluation.  So on one GPU with full-batch iterations update()
    # runs the actor's 80 iterations, hands the critic's 80 to a second HIP stream BEHIND them and
    # returns: the critic's kernels (on the compute units the resident collect kernel leaves, less 16
    # to spare: 219 of 256 at 256 workers) run while the host drives the next rollout, whose collect loop is latency-
    # bound and leaves the GPU idle.  Everything that reads the critic waits first (settle():
    # the next update, save / load, close, `last_infos`, the logger's dump — and the current stream
    # itself waits, so parameters read through torch are final), and what the critic still reads is
    # kept out of the rollout's way: the next rollout writes its observations to a spare buffer
    # (swapped in when it starts), the normaliser's mean / std (updated in place after the update,
    # a2c.py:126-127) are snapshot.  TONIC_AMD_CRITIC_OVERLAP=0: the interleaved launches of enqueue_update.
    OVERLAP_BLOCKS = None       # (developer override of the critic's workgroups per launch)
    MIN_OVERLAP_BLOCKS = 128    # below this width the critic's chain does not go under a rollout

    def _critic_blocks(self):
        """Workgroups of the critic's launches while a rollout is collected: what the resident
        collect kernel leaves of the 256 compute units (one workgroup per 16 workers + 4 copy + 1
        record, and a few to spare)."""
        if self.OVERLAP_BLOCKS is not None:
            return int(self.OVERLAP_BLOCKS)
        return min(256, max(8, 256 - self._collect_workgroups() - 16))

    def _critic_width(self):
        """`max_workgroups` of the critic's grad launches in this update: what they get under a
        rollout whenever the configuration is one whose critic MAY run there — whether or not it
        does (TONIC_AMD_CRITIC_OVERLAP) — so that the switch moves the critic's iterations in time
        and changes no bit of what they compute; the kernel's own width otherwise."""
        return self._critic_blocks() if self._overlap_eligible() else 0

    def _collect_workgroups(self):
        workers = getattr(self.replay, 'num_workers', None) or 0
        return (workers + 15) // 16 + 5

    def _overlap(self):
        return os.environ.get('TONIC_AMD_CRITIC_OVERLAP', '1') != '0' and self._overlap_eligible()

    def _overlap_eligible(self):
        """Everything but the switch: the updates of such a configuration give the critic's launches
        the width they have under a rollout (`_critic_blocks`) whether or not they run there, so the
        switch changes WHEN the critic's iterations run and not one bit of what they compute."""
        return (self.replay.batch_size is None
                and not self.actor_updater.stock and not self.critic_updater.stock
                and self.observation_size <= 32 and self.action_size <= 8
                # the collect kernel must leave the critic at least half of the chip: with thousands of
                # workers per GPU (cfg 5 on one GPU: 645 collect workgroups) a chain squeezed into the
                # few units left over would run many times longer than the rollout it hides under —
                # such configurations keep the full-width interleaved launches
                and self._critic_blocks() >= self.MIN_OVERLAP_BLOCKS
                and getattr(self, '_collector', None) is not None
                # only behind a rollout that came through the collector (it binds the Segment's
                # buffers anew every rollout; anybody else — rollout.DeviceRollout's captured graph —
                # may hold their addresses, and the overlap swaps the observation buffer)
                and getattr(self, '_host_rollout', False))

    @property
    def last_infos(self):
        self.settle()
        return self._last_infos

    @last_infos.setter
    def last_infos(self, value):
        self._last_infos = value

    def settle(self):
        """Waits for the critic iterations a previous update left running and logs their rows."""
        self._open_gate()
        pending = getattr(self, '_critic_pending', None)
        if pending is None:
            return
        self._critic_pending = None
        torch.cuda.current_stream().wait_event(pending['done'])   # whoever reads the critic next is behind it
        rows = pending['infos'][1].cpu().numpy()
        self.critic_chain_ms = pending['clock'].elapsed_time(pending['done'])    # (bench.py reports it)
        log_ppo_critic_rows(rows)
        logger.store('critic/iterations', len(rows))
        if getattr(self, '_last_infos', None) is not None:
            self._last_infos[1] = rows

    def save(self, path):
        self.settle()
        super().save(path)

    def load(self, path):
        self.settle()
        super().load(path)

    def close(self):
        self.settle()
        super().close()

    def _update(self):
        if not self._overlap():
            self.settle()
            infos = self.enqueue_update().cpu().numpy()          # the only sync of the update
            parallel.check_one_shot()
            log_ppo_update(infos)
            self.last_infos = infos
            if self.model.observation_normalizer:
                self.model.observation_normalizer.update()
            return
        self.settle()
        replay, actor, critic = self.replay, self.actor_updater, self.critic_updater
        critic.max_workgroups = self._critic_width()
        values, next_values = self._evaluate()
        replay.compute_returns(values, next_values)
        updates = replay.updates_per_get()
        infos = torch.zeros(2, updates, updaters.INFO_WIDTH, device=self.device)
        actor.reset_stop()
        buffers = replay.buffers
        obs, act, raw_adv, log_probs, returns = (
            replays.flatten_batch(buffers[k]) for k in replays.segments.LEARNER_KEYS)
        replay.index = 0
        n = obs.shape[0]
        first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        first.record()
        for it in range(updates):
            actor.enqueue_grad(obs, act, raw_adv, replay.adv_stats, log_probs)
            actor.enqueue_step(n, replay.adv_stats, infos[0, it])
        last.record()
        # the critic's iterations: same inputs, the normaliser as it is NOW, their own stream
        snapshot = tuple(t.clone() for t in critic.norm_tensors())
        normaliser_done = False
        if parallel.exchanging() and self.model.observation_normalizer:
            # Several ranks: the normaliser's update is a collective + a read-back on THIS stream; issued
            # after the critic's chain it would queue behind the chain's 80 all-reduces on the
            # communicator (one rank's host blocked for the whole chain).  It depends on neither
            # network (a2c.py:126-127 merges the rollout's recorded sums), the chain reads the snapshot:
            # it goes first.
            self.model.observation_normalizer.update()
            normaliser_done = True
        ready = torch.cuda.Event()
        ready.record()
        if getattr(self, '_critic_stream', None) is None:
            self._critic_stream = _side_stream('critic')
            logger.before_dump(self, 'settle')
            self._guard_critic_readers()
        side = self._critic_stream
        side.wait_event(ready)
        clock, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            self._arm_gate(side)
            clock.record(side)
            for it in range(updates):
                critic.enqueue_grad(obs, returns, norm=snapshot)
                critic.enqueue_step(n, infos[1, it])
            done.record(side)
        # (Enqueued HERE, while the GPU is busy with the actor's chain and the host has nothing else
        #  to do.  Round 4 also tried it late — from agent.update at the Segment row from which the
        #  chain just finishes before the rollout does, so that the next update's first launches find
        #  the chip at its working clock (profiles/r04_clock_ramp.md): the 160 enqueues then sit on the
        #  host-bound collect loop's critical path, 82.3 against 80.9 ms per step.)
        self._critic_pending = dict(done=done, clock=clock, infos=infos, keep=(obs, returns, snapshot))
        rows = infos[0].cpu().numpy()                    # waits for the actor's iterations only
        self.actor_chain_ms = first.elapsed_time(last)   # (bench.py reports it)
        parallel.check_one_shot()
        logger.store('actor/iterations', log_ppo_actor_rows(rows))
        self._last_infos = np.stack([rows, np.zeros_like(rows)])
        if self.model.observation_normalizer and not normaliser_done:
            self.model.observation_normalizer.update()
        # What the next rollout depends on — the actor, the normaliser — is on the current stream up
        # to here: a stream of its own carries that point to the collector (step() -> begin_rollout),
        # which also moves the rollout's observations to the spare buffer.  Everything ELSE that
        # follows on the current stream waits for the critic's iterations as well, and whoever reads
        # the critic through torch — state_dict(), a checkpoint, a forward pass of the module — goes
        # through settle() first (_guard_critic_readers), so the critic is read final, as after the
        # reference's update.
        if getattr(self, '_rollout_marker', None) is None:
            self._rollout_marker = _side_stream('rollout marker')
        ordered = torch.cuda.Event()
        ordered.record()
        self._rollout_marker.wait_event(ordered)
        self._rollout_behind = self._rollout_marker
        if self._gate_row is None:
            torch.cuda.current_stream().wait_event(done)
        # (A chain parked behind a gate is NOT put in front of the current stream: the gate opens late in
        #  the next rollout, and user work on this stream — a `current_stream().synchronize()` after
        #  agent.update — would stall for most of a rollout.  Whoever reads the critic is ordered behind
        #  the chain by settle(): the package's own readers, a forward pass of the module, state_dict().)
        self._rollout_started = time.perf_counter()

    # The critic's launches are enqueued above, while the host has nothing else to do, but they need not
    # START there: behind a phase in which the device is lightly loaded (a rollout keeps a few compute
    # units polling) the next update's first launches run 10 - 25 % slower until the device's power
    # management has followed the load (~25 ms: the whole actor chain, profiles/r04_clock_ramp.md).  A gate
    # (tonic_stream_gate: one polling wave) holds the chain back until the row of the NEXT rollout from
    # which it just finishes before that rollout does — the host opens it with a plain store from
    # agent.update — so the update that follows starts on a device at its working point.  Timing only;
    # whoever needs the critic (settle), the current stream (test_step) or a rollout slower than the
    # last one (the gate's own limit) opens it as well.  TONIC_AMD_CRITIC_GATE=0: the chain starts at once.
    GATE_MARGIN_MS = 1.0
    GATE_ROLLOUT_US = 100e3

    def _arm_gate(self, side):
        self._gate_row = None
        chain_ms, step_us = getattr(self, 'critic_chain_ms', None), getattr(self, '_rollout_step_us', None)
        if os.environ.get('TONIC_AMD_CRITIC_GATE', '1') == '0' or not chain_ms or not step_us:
            return
        if parallel.exchanging():
            # The chain's iterations all-reduce their gradient sums: a collective issued on the main
            # stream meanwhile (the normaliser's statistics) queues BEHIND them on the communicator,
            # and with a host-synchronous backend the first of them blocks the host — the only party
            # that can open a gate.  Ranks that exchange never hold the chain back.
            return
        rows = self.replay.max_size
        row = rows - int(np.ceil((1.1 * chain_ms + self.GATE_MARGIN_MS) * 1e3 / step_us))
        if row <= 0 or rows * step_us > self.GATE_ROLLOUT_US:
            # the chain needs the whole rollout / a rollout much longer than the device's ramp: nothing to
            # gain, and a polling wave must not sit in a hardware queue for long (HIP streams share a
            # handful of them: whatever another stream launches into that queue waits behind the gate)
            return
        if getattr(self, '_gate_word', None) is None:
            self._gate_word = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._gate_view = self._gate_word.numpy()
            self._gate_ticket = 0
        self._gate_ticket += 1
        # (its own limit: twice the time the host needs to that row — nobody who synchronises the device
        #  without asking the agent first waits longer than that)
        limit = max(0.01, 2.0 * row * step_us * 1e-6 + 0.005)
        _lib.check(_lib.load().tonic_stream_gate(self._gate_word.data_ptr(), self._gate_ticket, limit,
                                                 side.cuda_stream), 'stream gate')
        self._gate_row = row
        _ARMED_GATES.add(self)

    def _open_gate(self):
        # The critic stream is shared by the agents of a process (_side_stream): whatever another
        # agent has parked there holds THIS agent's chain too, so every armed gate is opened.
        for agent in list(_ARMED_GATES):
            if agent._gate_row is not None:
                agent._gate_view[0] = agent._gate_ticket
                agent._gate_row = None
        _ARMED_GATES.clear()

    def _guard_critic_readers(self):
        """Whoever reads the critic through torch — a forward pass of the module, state_dict() —
        finds it final: both settle first (the package's own readers call settle() themselves)."""
        def before(*args, **kwargs):
            self.settle()
        self.model.critic.register_forward_pre_hook(before)
        self.model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.settle())


def log_ppo_update(infos):
    """The keys ppo.py:46-67 logs for one learner update, from the statistics rows the device
    wrote: infos[0] = actor rows {loss, kl, entropy, clip_fraction, std, stop, ran}, one per
    iteration that ran before the KL stop; infos[1] = critic rows {loss, v}."""
    ran = log_ppo_actor_rows(infos[0])
    log_ppo_critic_rows(infos[1])
    logger.store('actor/iterations', ran)
    logger.store('critic/iterations', len(infos[1]))


def log_ppo_actor_rows(rows):
    """Stores the rows of the iterations that ran before the KL stop; returns how many did."""
    actor_rows = rows[rows[:, 6] > 0]
    for row in actor_rows:
        for i, key in enumerate(updaters.ACTOR_INFO):
            value = row[i] > 0.5 if key == 'stop' else row[i]
            logger.store('actor/' + key, value)
    return len(actor_rows)


def log_ppo_critic_rows(rows):
    for row in rows:
        logger.store('critic/loss', row[0])
        logger.store('critic/v', row[1])      # mean of the value batch (log-equivalent)


# ----------------------------------------------------------------- off-policy (SAC / TD3)

def shard_noise(eps, positions, counts, global_batch):
    """The rows of the GLOBAL noise draws that belong to this rank's part of each batch.
    eps [iterations, draws, S * B, A]: S draws per state of the global batch, sample-major (row
    s * B + m; S = 1 for everything but MPO).  positions[it, :counts[it]] = the batch positions m
    this rank owns.  Returns the same shape with row s * c + j = eps row s * B + positions[j]
    (c = counts[it]) in front and zeros behind: what the kernels see as a batch of c states."""
    local = np.zeros_like(eps)
    draws, per_sample = eps.shape[1], eps.shape[2] // global_batch
    for it in range(eps.shape[0]):
        c = counts[it]
        width = eps.shape[3]
        kept = eps[it].reshape(draws, per_sample, global_batch, width)[:, :, positions[it, :c]]
        local[it, :, :per_sample * c] = kept.reshape(draws, per_sample * c, width)   # (c may be 0)
    return local


def _twin_model(head):
    return models.ActorTwinCriticWithTargets(
        actor=models.Actor(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=head),
        critic=models.Critic(
            encoder=models.ObservationActionEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.ValueHead()),
        observation_normalizer=normalizers.MeanStd())


def _ddpg_model():
    """tonic/torch/agents/ddpg.py:7-17."""
    return models.ActorCriticWithTargets(
        actor=models.Actor(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.DeterministicPolicyHead()),
        critic=models.Critic(
            encoder=models.ObservationActionEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.ValueHead()),
        observation_normalizer=normalizers.MeanStd())


class DDPG(Agent):
    """tonic/torch/agents/ddpg.py:20-112: acting / storing / update scheduling; TD3 and SAC build
    on this class like in the reference."""

    policy_kind = 0          # tonic_policy_forward kind used by `_policy`

    def __init__(self, model=None, replay=None, exploration=None, actor_updater=None,
                 critic_updater=None):
        self.model = model or _ddpg_model()
        self.replay = replay or replays.Buffer()
        self.exploration = exploration or explorations.NormalActionNoise()
        self.actor_updater = actor_updater or updaters.DeterministicPolicyGradient()
        self.critic_updater = critic_updater or updaters.DeterministicQLearning()

    def initialize(self, observation_space, action_space, seed=None):
        super().initialize(seed=seed)
        self.device = _device()
        if not getattr(self, '_holds_affinity', False):
            self._holds_affinity = True
            parallel.hold_affinity()
        self.lib = _lib.load()
        self.model.initialize(observation_space, action_space)      # CPU init: seed parity
        self.model.pack(self.device)
        if self.model.observation_normalizer:
            self.model.observation_normalizer.attach(self.device)
        self.replay.initialize(seed, device=self.device)
        self.exploration.initialize(self._policy, action_space, seed)
        self.actor_updater.initialize(self.model)
        self.critic_updater.initialize(self.model)
        self.observation_size = observation_space.shape[0]
        self.action_size = action_space.shape[0]
        self.hidden = self.critic_updater.hidden
        # (the torch generator also feeds the update's GLOBAL noise stream here: it stays in step)
        self._replicate([self.model.flat_online, self.model.flat_target], own_noise=False)
        self._workers = None
        self._policy_io = {}
        self._q_blocks = {}
        self._q_last = None
        self._q_act = _lib.hot('tonic_collector_q_act')      # (per environment step: the vectorcall shim if built)
        self._actor_images = None             # the acting launch's weight images of the actor (tonic_collector_q_act)
        self._actor_images_stale = True
        self._actor_images_version = -1
        self._graph, self._static_key = None, None       # a re-initialised agent re-captures
        self._slots, self._slot_index = {}, 0

    def close(self):
        """Stores what is still deferred; gives back what the agent holds beyond its tensors (the process's CPU
        binding: parallel.bind_near_gpu)."""
        self.settle()
        if getattr(self, '_holds_affinity', False):
            self._holds_affinity = False
            parallel.release_affinity()

    # ------------------------------------------------------------------ acting
    def _block_of(self, observations, kind):
        """The collector block these observations live in when the policy can act on it in place
        (tonic_collector_q_act: the environments of tonic_amd.environments hand out views of their shared block;
        plain torsos the fused forward holds; one process per block's GPU handle) — None: the staged copies."""
        cached = self._q_last                      # (three look-ups per loop iteration: the same view each time)
        if cached is not None and cached[0] is observations and cached[1] == kind:
            return cached[2]
        state = self._find_block(observations, kind)
        if isinstance(observations, np.ndarray):
            self._q_last = (observations, kind, state)
        return state

    def _find_block(self, observations, kind):
        if kind not in (0, 1) or self.hidden is None or os.environ.get('TONIC_AMD_Q_BLOCK', '1') == '0':
            return None
        if self._actor_images is None:          # (once: does the fused forward on weight images serve this policy?)
            need = self.lib.tonic_mlp_actor_image_bytes(self.observation_size, self.hidden, self.action_size,
                                                        2 if kind == 1 else 1)
            self._actor_images = torch.zeros(need, dtype=torch.uint8, device=self.device) if need > 0 else False
            self._actor_images_stale = True
        if self._actor_images is False:
            return None
        if not isinstance(observations, np.ndarray):
            return None
        block = Block.owner_of(observations)
        if block is None:
            return None
        state = self._q_blocks.get(id(block))
        if state is None:
            W = block.workers
            need = self.lib.tonic_offpolicy_workspace_bytes(W, self.observation_size, self.action_size, self.hidden)
            state = dict(block=block, collector=Collector.for_block(block, 0),
                         workspace=torch.empty(need, dtype=torch.uint8, device=self.device),
                         rows=[torch.zeros(W, self.observation_size, device=self.device) for _ in range(2)],
                         turn=0, deferred=None, rows_of=None, actions_of=None, store_pending=False, usable=True)
            # the block's fields as the GPU sees them (page-locked by the collector): the store reads them in place
            def mapped(view, shape):
                address = self.lib.tonic_host_device_pointer(view.ctypes.data)
                return _DevicePointer(address, shape) if address else None
            W, O, A = block.workers, self.observation_size, self.action_size
            state['fields'] = dict(actions=mapped(block.actions, (W, A)),
                                   next_observations=mapped(block.next_observations, (W, O)),
                                   rewards=mapped(block.rewards, (W,)), resets=mapped(block.resets, (W,)),
                                   terminations=mapped(block.terminations, (W,)))
            if any(v is None for v in state['fields'].values()):
                state['usable'] = False
            self._q_blocks[id(block)] = state
        return state if state['usable'] else None

    def _act_on_block(self, state, kind, stochastic):
        """One launch on the environment's block, no copies: ddpg.py:45-52 / sac.py:40-51 (tonic_collector_q_act)."""
        block, collector = state['block'], state['collector']
        W = block.workers
        if stochastic:          # Normal.sample() of sac.py:43 == loc + scale * randn
            np.copyto(block.eps[0], self._randn(W, self.action_size).numpy())
        workspace = state['workspace']
        # the images follow the float32 parameters: rebuilt after every learner update (raw-pointer writes: the
        # flag) and whenever torch has written the flat block in place since (load_state_dict, an optimizer of the
        # caller's: the tensor's version counter, which every in-place operation on a view of it advances)
        flat = self.model.flat_online
        stale = self._actor_images_stale or flat._version != self._actor_images_version
        self._actor_images_version = flat._version
        # the transition of the step before (reserved by update(), its sources still in the block) rides in this
        # launch: one more workgroup stores it while the tiles compute these actions
        deferred, store = state['deferred'], None
        if deferred is not None:
            held = state.get('store_struct')        # (one tonic_q_store_t per block: two fields change per step)
            if held is None or state.get('store_struct_of') is not self.replay.buffers:
                held = state['store_struct'] = self.replay.store_arguments(
                    0, state['rows'][0], self.model.observation_normalizer)
                state['store_struct_of'] = self.replay.buffers
                state['store_struct_rows'] = [_lib.ptr(r) for r in state['rows']]
            held.row = deferred['row']
            held.d_observations = state['store_struct_rows'][deferred['turn']]
            store = ctypes.addressof(held)
        turn = state['turn'] ^ 1          # (these rows' device copy: the buffer the pending store does not read)
        status = self._q_act(
            collector.handle, _lib.ptr(self.model.flat_actor.flat), _lib.ptr(self._actor_images),
            int(stale), kind, self.hidden, 0 if stochastic else -1,
            _lib.ptr(state['rows'][turn]), store, _lib.ptr(workspace), workspace.numel(), _lib.current_stream())
        if status != 0:
            _lib.check(status, 'tonic_collector_q_act')
        self._actor_images_stale = False
        collector.wait_actions()          # (every completion word, the store's included)
        state['deferred'], state['store_pending'] = None, False
        state['turn'], state['rows_of'] = turn, block
        return block.eps[1].copy()

    def _flush_store(self, state):
        """A reserved transition goes out as a launch of its own (no acting launch will carry it in time)."""
        deferred = state['deferred']
        if deferred is None:
            return
        self.replay.store_at(deferred['row'], self.model.observation_normalizer,
                             observations=state['rows'][deferred['turn']], **state['fields'])
        state['deferred'], state['store_pending'] = None, True

    def settle(self):
        """Everything this agent has deferred is on the device: reserved transitions stored, their launches through
        (readers of `replay.buffers` between two steps call this; `close` does)."""
        for state in getattr(self, '_q_blocks', {}).values():
            self._flush_store(state)
            if state['store_pending']:
                torch.cuda.current_stream().synchronize()
                state['store_pending'] = False

    def _forward_policy(self, observations, kind, stochastic):
        state = self._block_of(observations, kind)
        for other in self._q_blocks.values():
            # a transition reserved on ANOTHER block (test episodes between training steps, a second environment)
            # does not wait for that block's next acting launch
            if other is not state and other['deferred'] is not None:
                self._flush_store(other)
        if state is not None:
            actions = self._act_on_block(state, kind, stochastic)
            if actions is not None:
                return actions
        observations = np.asarray(observations, np.float32)
        W = observations.shape[0]
        io = self._policy_io.get(W)
        if io is None:
            need = 16 if self.hidden is None else self.lib.tonic_offpolicy_workspace_bytes(
                W, self.observation_size, self.action_size, self.hidden)
            io = (_Staging([('observations', (W, self.observation_size)),
                            ('eps', (W, self.action_size))], self.device),
                  _Staging([('actions', (W, self.action_size))], self.device),
                  torch.empty(need, dtype=torch.uint8, device=self.device))
            self._policy_io[W] = io
        stage_in, stage_out, workspace = io
        stage_in.host_view('observations')[:] = observations
        if stochastic:      # Normal.sample() of sac.py:43 == loc + scale * randn
            stage_in.host_view('eps')[:] = self._randn(W, self.action_size).numpy()
        stage_in.upload()
        if self.hidden is None:
            # any MLP(sizes, activation): the policy's forward as stock torch operators (updaters.
            # _StockTorch); kind 0 tanh head, 1 squashed Gaussian (greedy: its loc), 2 Gaussian
            with torch.no_grad():
                out = self.model.actor(
                    stage_in.device_view('observations').to(self.device, non_blocking=True))
                eps = stage_in.device_view('eps').to(self.device, non_blocking=True)
                if kind == 0:
                    actions = out
                elif kind == 1:
                    normal = out._distribution
                    actions = torch.tanh(normal.mean + normal.stddev * eps) if stochastic else out.loc
                else:
                    actions = out.loc + out.scale * eps if stochastic else out.loc
                stage_out.device_view('actions').copy_(actions, non_blocking=True)
            stage_out.download()
            stage_out.mark()
            stage_in.done = stage_out.done
            stage_out.wait()
            return stage_out.host_view('actions').copy()
        p = _lib.ptr
        _lib.check(self.lib.tonic_policy_forward(
            p(self.model.flat_actor.flat), p(stage_in.device_view('observations')),
            p(stage_in.device_view('eps')) if stochastic else None,
            p(stage_out.device_view('actions')), kind, W, self.observation_size, self.hidden,
            self.action_size, p(workspace), workspace.numel(), _lib.current_stream()),
            'tonic_policy_forward')
        stage_out.download()
        stage_out.mark()
        stage_in.done = stage_out.done
        stage_out.wait()
        return stage_out.host_view('actions').copy()

    def _greedy_actions(self, observations):
        return self._forward_policy(observations, self.policy_kind, False)

    def _policy(self, observations):
        return self._greedy_actions(observations)

    def step(self, observations, steps):
        for state in self._q_blocks.values():
            state['rows_of'] = state['actions_of'] = None
        actions = self.exploration(observations, steps)
        self.last_observations = observations.copy()
        self.last_actions = actions.copy()
        self._stepped_on = observations
        state = self._block_of(observations, self.policy_kind)
        if state is not None:
            block = state['block']
            if state['rows_of'] is not block:
                # the policy sat this step out (warm-up: uniform actions): no acting launch carried the reserved
                # transition or was waited for behind a store launch — they read the block in place and must be
                # through before the environment overwrites it
                self._flush_store(state)
                if state['store_pending']:
                    torch.cuda.current_stream().synchronize()
            state['store_pending'] = False
            # the executed actions go into the environment's block (where the store launch reads them); policy
            # actions (float32) are handed out as the block's own view — value-identical, and the environments of
            # tonic_amd.environments take their one-call step then — the warm-up's float64 draws as they are
            np.copyto(block.actions, actions)
            state['actions_of'] = block
            if actions.dtype == np.float32:
                return block.out_actions
        return actions

    def test_step(self, observations, steps):
        return self._greedy_actions(observations)

    # ---------------------------------------------------------------- learning
    def _store_from_block(self, observations, rewards):
        """The transition straight from the environment's block (Buffer.store, buffers.py:33-56): next observations,
        outcome and the executed actions are read by the store launch IN PLACE (page-locked block), the observations
        from the device copy the acting launch made — or, when the policy did not act on this step (warm-up), from one
        staged copy.  False: not this step's layout (foreign arrays, another block): the staged path."""
        block = Block.owner_of(getattr(self, '_stepped_on', None)) if isinstance(
            getattr(self, '_stepped_on', None), np.ndarray) else None
        if block is None or observations is not block.out_next_observations or rewards is not block.out_rewards:
            return False
        state = self._block_of(self._stepped_on, self.policy_kind)
        if state is None or state.get('actions_of') is not block:      # (step() put the executed actions there)
            return False
        normalizer = self.model.observation_normalizer
        if state['rows_of'] is block and self.replay.buffers is not None and self.replay.return_steps == 1:
            # the usual step: the row is reserved now (Buffer bookkeeping, buffers.py:54-56) and written by the next
            # acting launch, which reads the block before the environment's next step can touch it
            state['deferred'] = dict(row=self.replay.reserve_row(normalizer), turn=state['turn'])
            return True
        rows = state['rows'][state['turn']]
        if state['rows_of'] is not block:       # (warm-up: uniform actions, the policy never saw these rows)
            staged = state.setdefault('staged_rows', torch.zeros(
                block.workers, self.observation_size, dtype=torch.float32).pin_memory())
            torch.cuda.current_stream().synchronize()        # (the previous store may still read it)
            staged.numpy()[:] = self.last_observations
            rows.copy_(staged, non_blocking=True)
        self.replay.store(normalizer=normalizer, observations=rows, **state['fields'])
        state['store_pending'] = True
        return True

    def update(self, observations, rewards, resets, terminations, steps):
        if self._store_from_block(observations, rewards):
            if self.model.return_normalizer:
                raise NotImplementedError('return normalisers are not supported')
            if self.replay.ready(steps):
                for state in self._q_blocks.values():
                    self._flush_store(state)         # the update samples this transition too
                self._update(steps)          # (ends with a read-back: the store launches are through)
                for state in self._q_blocks.values():
                    state['store_pending'] = False
            self.exploration.update(resets)
            return
        W = len(rewards)
        if self._workers != W:
            self._workers = W
            O, A = self.observation_size, self.action_size
            fields = [('observations', (W, O)), ('actions', (W, A)), ('next_observations', (W, O)),
                      ('rewards', (W,)), ('resets', (W,)), ('terminations', (W,))]
            # two pinned blocks used in turn: the copy of step t may still be in flight while the
            # host fills the block of step t+1
            self._transitions = (_Staging(fields, self.device), _Staging(fields, self.device))
            self._transition_turn = 0
        self._transition_turn ^= 1
        stage = self._transitions[self._transition_turn].writable()
        stage.host_view('observations')[:] = self.last_observations
        stage.host_view('actions')[:] = self.last_actions          # float64 warm-up -> float32
        stage.host_view('next_observations')[:] = observations
        stage.host_view('rewards')[:] = rewards
        stage.host_view('resets')[:] = resets
        stage.host_view('terminations')[:] = terminations
        stage.upload()
        self.replay.store(
            normalizer=self.model.observation_normalizer,
            **{k: stage.device_view(k) for k in ('observations', 'actions', 'next_observations',
                                                 'rewards', 'resets', 'terminations')})
        stage.mark()                  # the store kernel(s) read the pinned fields in place
        if self.model.return_normalizer:
            raise NotImplementedError('return normalisers are not supported')
        if self.replay.ready(steps):
            self._update(steps)
        self.exploration.update(resets)

    def _actor_due(self, iteration):
        return True                       # ddpg.py:105-112: actor + targets every iteration

    def enqueue_update(self, indices, eps, graph=None):
        """Enqueues `indices.shape[0]` learner iterations (no host sync).  indices: int64
        [iterations, B] host array; eps: float32 [iterations, draws, B, A] host array with the
        standard-normal draws in the order the reference consumes them.  With `graph` (default:
        on unless TONIC_AMD_NO_GRAPH=1) the launch sequence — gather, the fused forward / backward /
        grouped weight-gradient launches, Adam + polyak: 13 per SAC iteration — is captured once into a hipGraph
        reading fixed index / noise buffers and replayed on later calls."""
        iterations, global_batch = indices.shape
        self._actor_images_stale = True          # (the acting launch rebuilds its weight images after an update)
        world = self.critic_updater.world_size
        counts = None
        if world > 1:
            # every rank draws the same GLOBAL index / noise stream and keeps the
            # samples whose worker column lives in its shard (binomial split, mean B / world);
            # gradient SUMS are all-reduced and scaled by 1 / B_global, so the update equals the
            # single-process one on the global buffer.
            indices, positions, counts = self.replay.shard_indices(indices)
            eps = shard_noise(eps, positions, counts, global_batch)
        # everything a captured graph bakes in: shapes, the replay's storage, the updaters'
        # hyper-parameters and schedules — a change of any of them re-captures
        key = (iterations, tuple(eps.shape), self.replay.buffers['observations'].data_ptr(),
               self.replay.max_size, self._graph_signature())
        if getattr(self, '_static_key', None) != key:
            self._static_key = key
            self._static_indices = torch.zeros(indices.shape, dtype=torch.int64, device=self.device)
            self._static_eps = torch.zeros(eps.shape, dtype=torch.float32, device=self.device)
            self._infos = torch.zeros(2, iterations, updaters.INFO_WIDTH, device=self.device)
            self._graph = None
        self._static_indices.copy_(torch.as_tensor(indices), non_blocking=True)
        self._static_eps.copy_(torch.as_tensor(eps), non_blocking=True)
        if graph is None:
            graph = os.environ.get('TONIC_AMD_NO_GRAPH', '0') != '1'

        fused = self._fused_kind()
        # Several ranks / gradient clipping need the complete gradient sums between gradients and step: the
        # same chained launches in two halves around the exchange (tonic_q_iteration_t.phase), the steps
        # through the updaters' own `_step` (all-reduce, clip, Adam [+ polyak]) like the split entry points
        phased = fused is not None and self._fused_in_phases()
        self._phased_update = phased
        if phased:
            self._fused_workspace_for(max(indices.shape[1], 1))[:4].zero_()   # the update's failure word
        if fused is not None and not phased:
            # the optimizer steps' float64 constants, formed here like the reference's Python
            # floats: [iteration, {critic, actor}, {step_size, bias_correction2_sqrt}]
            table = np.zeros((iterations, 2, 2), np.float32)
            critic, actor = self.critic_updater, self.actor_updater
            if not getattr(self, '_step_mirror_valid', False):
                # the host mirrors of the device's step counters are only kept by THIS path: after
                # an update on the split entry points (whose captured graphs step the device counter
                # on every replay), an exception on the way to the launches or a skipped step, the
                # counters on the device are the authority (one read-back, then none again)
                critic.steps_enqueued = int(critic.state[0])
                actor.steps_enqueued = int(actor.state[0])
            self._step_mirror_valid = False          # until this update's launches are out
            for it in range(iterations):
                critic.steps_enqueued += 1
                table[it, 0] = updaters.adam_step_constants(critic.hyper, critic.steps_enqueued)
                if self._actor_due(it):
                    actor.steps_enqueued += 1
                    table[it, 1] = updaters.adam_step_constants(actor.hyper, actor.steps_enqueued)
            if getattr(self, '_static_adam', None) is None or self._static_adam.shape != table.shape:
                self._static_adam = torch.zeros(table.shape, device=self.device)
                self._graph = None
            self._static_adam.copy_(torch.as_tensor(table), non_blocking=True)

        def enqueue():
            self._infos.zero_()
            draws = self._static_eps.shape[1]
            per_sample = self._static_eps.shape[2] // global_batch     # (MPO: num_samples rows each)
            # the store does not change during an update: the batches of ALL its iterations are
            # gathered by one launch (the rows a rank does not own are padding, never read)
            batches = self.replay.gather_many(self._static_indices)
            if fused is not None and not phased and counts is None:
                self._enqueue_fused_iterations(fused, batches, iterations)
                return
            for it in range(iterations):
                c = global_batch if counts is None else int(counts[it])
                n_global = None if counts is None else global_batch
                if c > 0:
                    batch = {k: v[it, :c] for k, v in batches.items()}
                if phased:
                    self._enqueue_phased(fused, batch if c > 0 else None, it, self._actor_due(it), n_global)
                    continue
                if fused is not None:
                    self._enqueue_fused(fused, batch, it, self._actor_due(it))
                    continue
                if c > 0:
                    self.critic_updater.enqueue(batch, self._static_eps[it, 0, :c * per_sample],
                                                self._infos[0, it], n_global)
                else:
                    self.critic_updater.enqueue_empty(self._infos[0, it], n_global)
                if self._actor_due(it):
                    actor_eps = self._static_eps[it, 1, :c * per_sample] if draws > 1 else None
                    # update_targets() (ddpg.py:112) rides in the actor's optimizer launch
                    targets = (self.model.flat_target, self.model.flat_online, 0,
                               self.model.target_coeff)
                    if c > 0:
                        self._enqueue_actor(batch['observations'], actor_eps, it, n_global, targets)
                    else:
                        self._enqueue_actor(None, None, it, n_global, targets)

        stock = self.critic_updater.stock or self.actor_updater.stock    # (autograd: no capture)
        if not graph or stock or parallel.exchanging():      # collectives sit between the kernels
            enqueue()
            self._step_mirror_valid = fused is not None and not phased
            return self._infos
        if self._graph is None:
            batch_size = indices.shape[1]
            self.critic_updater._offpolicy_workspace(batch_size)    # allocate outside the capture
            self.actor_updater._offpolicy_workspace(batch_size)
            self.replay.gather_many(self._static_indices)
            if fused is not None:
                self._fused_workspace_for(batch_size)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with _lib.capturing(self._graph):
                enqueue()
        self._graph.replay()
        self._step_mirror_valid = fused is not None and not phased
        return self._infos

    def _enqueue_actor(self, observations, eps, iteration, n_global, targets):
        if observations is None:                    # this rank drew none of the global batch
            self.actor_updater.enqueue_empty(self._infos[1, iteration], n_global, targets)
            return
        self.actor_updater.enqueue(observations, eps, self._infos[1, iteration], n_global, targets)

    # -- the whole iteration through tonic_q_iteration (5 launches instead of 13)
    _FUSED = {updaters.DeterministicQLearning: (2, updaters.DeterministicPolicyGradient),
              updaters.TwinCriticDeterministicQLearning: (0, updaters.DeterministicPolicyGradient),
              updaters.TwinCriticSoftQLearning: (1, updaters.TwinCriticSoftDeterministicPolicyGradient)}

    def _fused_kind(self):
        """The `kind` of tonic_q_iteration when it serves this agent: the plain DDPG / TD3 / SAC
        updaters, one rank (several ranks all-reduce between gradients and step), no gradient
        clipping (needs the whole gradient before the step), shapes inside the fused kernels.
        TONIC_AMD_FUSED_ITERATION=0 keeps the split entry points (the tests compare the two)."""
        critic, actor = self.critic_updater, self.actor_updater
        kind, actor_class = self._FUSED.get(type(critic), (None, None))
        if kind is None or type(actor) is not actor_class or critic.stock or actor.stock:
            return None
        if os.environ.get('TONIC_AMD_FUSED_ITERATION', '1') == '0':
            return None
        if not self.lib.tonic_q_iteration_supported(self.observation_size, self.hidden,
                                                    self.action_size, 2 if kind == 1 else 1):
            return None
        needs_phases = parallel.exchanging() or critic.world_size > 1 or critic.gradient_clip > 0 \
            or actor.gradient_clip > 0
        if needs_phases and os.environ.get('TONIC_AMD_FUSED_PHASES', '1') == '0':
            return None
        return kind

    def _fused_in_phases(self):
        """The fused iteration in two halves, gradient sums only (tonic_q_iteration_t.phase 1 / 2): whenever
        something has to see the complete sums before the step — the exchange between ranks, a gradient-norm
        clip.  TONIC_AMD_FUSED_PHASES=0: the split entry points there, as before round 5."""
        critic, actor = self.critic_updater, self.actor_updater
        needed = parallel.exchanging() or critic.world_size > 1 or critic.gradient_clip > 0 \
            or actor.gradient_clip > 0
        return needed and os.environ.get('TONIC_AMD_FUSED_PHASES', '1') != '0'

    def _fused_workspace_for(self, batch_size):
        need = self.lib.tonic_q_iteration_workspace_bytes(batch_size, self.observation_size,
                                                          self.action_size, self.hidden)
        ws = getattr(self, '_fused_workspace', None)
        if ws is None or ws.numel() < need:
            # (zero-filled: its head holds the arrival words of the chained launches)
            self._fused_workspace = ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return ws

    def _enqueue_phased(self, kind, batch, iteration, actor_due, n_global):
        """One iteration as [policy passes + critic step + critics' weight gradients] -> exchange / clip + Adam
        -> [actor step + actor's weight gradients] -> exchange / clip + Adam + polyak: 3 + 1 + 2 + 1 launches
        around the two exchanges instead of the split entry points' 4 + 1 + 5 + 1."""
        critic, actor, model = self.critic_updater, self.actor_updater, self.model
        targets = (model.flat_target, model.flat_online, 0, model.target_coeff)
        if batch is None:                           # this rank drew none of the global batch
            critic.enqueue_empty(self._infos[0, iteration], n_global)
            if actor_due:
                self._enqueue_actor(None, None, iteration, n_global, targets)
            return
        B = batch['observations'].shape[0]
        n = n_global or B * critic.world_size
        self._enqueue_fused(kind, batch, iteration, actor_due, phase=1)
        critic._step(n, self._infos[0, iteration])
        if actor_due:
            self._enqueue_fused(kind, batch, iteration, actor_due, phase=2)
            actor._step(n, self._infos[1, iteration], targets=targets)

    def _enqueue_fused_iterations(self, kind, batches, iterations):
        """The fused iterations of one update call.  With delayed actor updates (td3.py:43-46) the policy passes
        (launch 1) of iteration k + 1 — they read the actor, the target actor, ITS batch and noise, none of which a
        critic step writes — ride as more workgroups in the critic-step launch of an iteration k that does not step
        the actor (tonic_q_iteration_t.ahead: the workspace's other set of launch-1 outputs; iteration k + 1 then
        starts behind them, stage 2).  At B = 100 a launch fills a ninth of the chip: one launch and its dispatch
        less per pair of iterations.  Same kernels' code on the same inputs: the same bits
        (TONIC_AMD_POLICY_AHEAD=0: every iteration by itself; the tests compare)."""
        due = [bool(self._actor_due(it)) for it in range(iterations)]
        B = batches['observations'].shape[1]
        nets = 1 if kind == 2 else 2
        ahead_on = os.environ.get('TONIC_AMD_POLICY_AHEAD', '1') != '0' and not all(due)
        args, rides = [], []             # rides[it]: iteration it + 1's policy passes run in it's critic step
        slot = 0
        for it in range(iterations):
            args.append(self._fused_arguments(kind, {k: v[it] for k, v in batches.items()}, it, due[it],
                                              stage=2 if it and rides[it - 1] else 0, slot=slot))
            rides.append(ahead_on and not due[it] and it + 1 < iterations and bool(
                self.lib.tonic_q_iteration_ahead_supported(B, self.observation_size, self.hidden, self.action_size,
                                                           nets, 2 if due[it + 1] else 1)))
            if rides[it]:
                slot ^= 1                # (the passes ahead write the other set: the next iteration's)
        for it in range(iterations):
            if rides[it]:
                args[it].ahead = ctypes.addressof(args[it + 1])      # (read during this call only; `args` lives on)
            _lib.check(self.lib.tonic_q_iteration(ctypes.byref(args[it]), _lib.current_stream()),
                       'tonic_q_iteration')

    def _enqueue_fused(self, kind, batch, iteration, actor_due, phase=0):
        _lib.check(self.lib.tonic_q_iteration(
            ctypes.byref(self._fused_arguments(kind, batch, iteration, actor_due, phase)), _lib.current_stream()),
            'tonic_q_iteration')

    def _fused_arguments(self, kind, batch, iteration, actor_due, phase=0, stage=0, slot=0):
        critic, actor, model, p = self.critic_updater, self.actor_updater, self.model, _lib.ptr
        B = batch['observations'].shape[0]
        ws = self._fused_workspace_for(B)
        mean, std = critic.norm_tensors()
        noise = getattr(critic, 'target_action_noise', None)
        eps = self._static_eps

        def optimizer(updater, info_row, constants):
            h = updater.hyper
            return _lib.QOptimizer(p(updater.grad_sums), p(updater.exp_avg), p(updater.exp_avg_sq),
                                   p(updater.state), p(info_row), p(constants), h['lr'],
                                   h['betas'][0], h['betas'][1], h['eps'])
        return _lib.QIteration(
            kind=kind, actor_due=int(bool(actor_due)), B=B, O=self.observation_size, H=self.hidden,
            A=self.action_size, global_batch=B,
            d_actor=p(model.flat_actor.flat), d_critics=p(model.flat_critics.flat),
            d_target_actor=p(model.flat_target_actor.flat),
            d_target_critics=p(model.flat_target_critics.flat),
            d_norm_mean=p(mean), d_norm_std=p(std), norm_clip=critic.norm_clip(),
            d_observations=p(batch['observations']), d_actions=p(batch['actions']),
            d_next_observations=p(batch['next_observations']), d_rewards=p(batch['rewards']),
            d_discounts=p(batch['discounts']),
            d_eps_critic=p(eps[iteration, 0]) if kind != 2 else None,
            d_eps_actor=p(eps[iteration, 1]) if kind == 1 else None,
            critic_entropy_coeff=float(getattr(critic, 'entropy_coeff', 0.0)),
            actor_entropy_coeff=float(getattr(actor, 'entropy_coeff', 0.0)),
            noise_scale=float(noise.scale if noise else 0.0),
            noise_clip=float(noise.clip if noise else 0.0),
            target_coeff=float(model.target_coeff),
            critic=optimizer(critic, self._infos[0, iteration],
                             self._static_adam[iteration, 0] if phase == 0 else None),
            actor=optimizer(actor, self._infos[1, iteration],
                            self._static_adam[iteration, 1] if phase == 0 else None),
            d_workspace=p(ws), workspace_bytes=ws.numel(), phase=phase, stage=stage, slot=slot,
            # the workspace's fp16x2 weight images follow the optimizer epilogues INSIDE an update call; between
            # calls anybody may have written parameters (load_state_dict, another path): rebuilt on iteration 0
            refresh_images=int(iteration == 0 or phase != 0))

    def _graph_signature(self):
        parts = []
        for updater in (self.actor_updater, self.critic_updater):
            hyper = updater.hyper
            noise = getattr(updater, 'target_action_noise', None)
            parts.append((hyper['lr'], hyper['betas'], hyper['eps'],
                          float(getattr(updater, 'entropy_coeff', 0.0)),
                          float(getattr(updater, 'gradient_clip', 0.0) or 0.0),
                          (noise.scale, noise.clip) if noise is not None else None))
        norm = self.model.observation_normalizer
        return (tuple(parts), float(self.model.target_coeff), getattr(self, 'delay_steps', 1),
                norm._mean.data_ptr() if norm is not None else 0)

    def _draw_noise(self, iterations):
        # DeterministicQLearning / DeterministicPolicyGradient draw nothing (one unused slot)
        return np.zeros((iterations, 1, self.replay.batch_size, self.action_size), np.float32)

    # -- an update call in chunks: the host draws chunk k + 1's index and noise streams while the GPU runs chunk k
    _SLOT_FIELDS = ('_static_key', '_static_indices', '_static_eps', '_infos', '_graph', '_static_adam')

    def _select_slot(self, index):
        """Two sets of everything a captured update graph reads (index / noise / step-constant buffers, the
        statistics rows, the graph itself): one is being replayed while the host fills the other."""
        current = getattr(self, '_slot_index', 0)
        if index == current:
            return
        slots = self.__dict__.setdefault('_slots', {})
        slots[current] = {f: getattr(self, f, None) for f in self._SLOT_FIELDS}
        chosen = slots.get(index, {})
        for f in self._SLOT_FIELDS:
            setattr(self, f, chosen.get(f))
        self._slot_index = index

    def _update_chunk(self, iterations):
        """How many iterations one graph of a chunked update holds (0: one graph for the whole call).  The streams
        are drawn in the reference's order either way (indices: the Buffer's RandomState, noise: torch's CPU
        generator, both only consumed by this call while it runs) — chunking only moves WHEN the host draws them:
        7.6 -> ~6 ms per SAC update call of 50 iterations, whose 2 ms of host draws ran in front of 5.4 ms of GPU
        work (profiles/r06_offpolicy_pmc.md).  Needs the fused iteration in a hipGraph on one rank; chunks hold a
        whole number of actor delays.  TONIC_AMD_UPDATE_CHUNK=0: off; =n: n iterations per chunk."""
        chunk = int(os.environ.get('TONIC_AMD_UPDATE_CHUNK', '10'))
        delay = int(getattr(self, 'delay_steps', 1) or 1)
        if (chunk <= 0 or iterations < 2 * chunk or iterations % chunk or chunk % delay
                or type(self)._update is not DDPG._update or type(self).enqueue_update is not DDPG.enqueue_update
                or os.environ.get('TONIC_AMD_NO_GRAPH', '0') == '1' or parallel.exchanging()
                or self.critic_updater.world_size > 1 or self._fused_kind() is None or self._fused_in_phases()):
            return 0
        return chunk

    def _update(self, steps):
        replay = self.replay
        chunk = self._update_chunk(replay.batch_iterations)
        if chunk:
            parts = []
            for k in range(replay.batch_iterations // chunk):
                indices = replay.sample_indices(chunk)         # (host: while the GPU runs the chunk before)
                eps = self._draw_noise(chunk)
                self._select_slot(k & 1)
                parts.append(self.enqueue_update(indices, eps).clone())
            self._select_slot(0)
            infos = torch.cat(parts, dim=1).cpu().numpy()
        else:
            indices = replay.sample_indices()
            eps = self._draw_noise(indices.shape[0])
            infos = self.enqueue_update(indices, eps).cpu().numpy()
        parallel.check_one_shot()
        if getattr(self, '_phased_update', False) and int(self._fused_workspace[:4].view(torch.int32)[0]):
            # (in phases the steps are the updaters' own launches: nothing held them back)
            raise _lib.TonicHipError(
                'a chained launch of this update gave up waiting for a peer workgroup (250 ms): its gradient '
                'sums were incomplete and have been stepped on — the parameters are no longer valid '
                '(TONIC_AMD_TUNING=q_chain=0 runs one launch per pass)')
        try:
            _check_chain(infos)
        except _lib.TonicHipError:
            self._step_mirror_valid = False          # skipped steps: the device's counters decide
            raise
        replay.last_steps = steps
        twin = hasattr(self.model, 'critic_2')
        for row in infos[0]:
            logger.store('critic/loss', row[0])
            if twin:
                logger.store('critic/q1', row[1])    # batch means (log-equivalent)
                logger.store('critic/q2', row[2])
            elif not getattr(self.critic_updater, 'atoms', 0):
                logger.store('critic/q', row[1])
        for row in infos[1][infos[1][:, 6] > 0]:
            logger.store('actor/loss', row[0])
        self.last_infos = infos
        if self.model.observation_normalizer:
            self.model.observation_normalizer.update()


def _check_chain(infos):
    """A chained launch whose workgroups waited 250 ms for a value of a peer that never came (a GPU
    shared with something that starves the launch, a lost workgroup) does not hang and does not step:
    the iteration's optimizer epilogues wrote nothing and marked their statistic rows (slot 7).  The
    parameters are intact; the update is incomplete, which is an error and not a log line."""
    gave_up = np.argwhere(infos[..., 7] > 0)
    if len(gave_up):
        raise _lib.TonicHipError(
            f'{len(gave_up)} optimizer steps of this update were skipped: a chained launch gave up '
            f'waiting for a peer workgroup (first: updater {gave_up[0][0]}, iteration '
            f'{gave_up[0][1]}); parameters, moments and targets are those of the last complete '
            'iteration (TONIC_AMD_TUNING=q_chain=0 runs one launch per pass)')


def _d4pg_model():
    """tonic/torch/agents/d4pg.py:7-18 (support for the control suite with 0.99 discount)."""
    return models.ActorCriticWithTargets(
        actor=models.Actor(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.DeterministicPolicyHead()),
        critic=models.Critic(
            encoder=models.ObservationActionEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU),
            head=models.DistributionalValueHead(-150., 150., 51)),
        observation_normalizer=normalizers.MeanStd())


class D4PG(DDPG):
    """tonic/torch/agents/d4pg.py:21-37: DDPG's acting / storing / scheduling with a categorical
    critic on 5-step returns (the HBM Buffer accumulates them, buffers.py:58-79)."""

    def __init__(self, model=None, replay=None, exploration=None, actor_updater=None,
                 critic_updater=None):
        super().__init__(
            model or _d4pg_model(), replay or replays.Buffer(return_steps=5), exploration,
            actor_updater or updaters.DistributionalDeterministicPolicyGradient(),
            critic_updater or updaters.DistributionalDeterministicQLearning())


def _mpo_model():
    """tonic/torch/agents/mpo.py:7-18."""
    return models.ActorCriticWithTargets(
        actor=models.Actor(
            encoder=models.ObservationEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.GaussianPolicyHead()),
        critic=models.Critic(
            encoder=models.ObservationActionEncoder(),
            torso=models.MLP((256, 256), torch.nn.ReLU), head=models.ValueHead()),
        observation_normalizer=normalizers.MeanStd())


class MPO(DDPG):
    """tonic/torch/agents/mpo.py:21-109: acts by sampling the Gaussian policy (no exploration
    object), stores 5-step returns, and per batch runs ExpectedSARSA, the MPO actor / dual step
    and the target update — DDPG's staging, HBM Buffer, hipGraph capture and schedule."""

    policy_kind = 2

    def __init__(self, model=None, replay=None, actor_updater=None, critic_updater=None):
        super().__init__(
            model or _mpo_model(), replay or replays.Buffer(return_steps=5),
            explorations.NoActionNoise(start_steps=0),
            actor_updater or updaters.MaximumAPosterioriPolicyOptimization(),
            critic_updater or updaters.ExpectedSARSA())

    def step(self, observations, steps):
        actions = self._forward_policy(observations, 2, True)       # mpo.py:38-46, 77-80
        self.last_observations = observations.copy()
        self.last_actions = actions.copy()
        return actions

    def _draw_noise(self, iterations):
        # per iteration: rsample((S,)) of ExpectedSARSA (critics.py:260), then sample((S,)) of the
        # actor step (actors.py:359) — [S, B, A] standard normals each, kept as [S * B, A]
        B, A = self.replay.batch_size, self.action_size
        S_c, S_a = self.critic_updater.num_samples, self.actor_updater.num_samples
        if S_c != S_a:
            raise NotImplementedError('ExpectedSARSA and MPO must draw the same number of samples')
        return np.stack([np.stack([torch.randn(S_c, B, A).numpy().reshape(S_c * B, A),
                                   torch.randn(S_a, B, A).numpy().reshape(S_a * B, A)])
                         for _ in range(iterations)])

    def enqueue_update(self, indices, eps, graph=None):
        width = 9 + 2 * self.action_size
        if getattr(self, '_mpo_stats', None) is None or self._mpo_stats.shape[0] != indices.shape[0]:
            self._mpo_stats = torch.zeros(indices.shape[0], width, device=self.device)
            self._graph = None
        return super().enqueue_update(indices, eps, graph)

    def _enqueue_actor(self, observations, eps, iteration, n_global, targets):
        if observations is None:                    # this rank drew none of the global batch
            self.actor_updater.enqueue_empty(self._infos[1, iteration], n_global, targets,
                                             stats_row=self._mpo_stats[iteration])
            return
        self.actor_updater.enqueue(observations, eps, self._infos[1, iteration], n_global, targets,
                                   stats_row=self._mpo_stats[iteration])

    def _update(self, steps):
        replay = self.replay
        indices = replay.sample_indices()
        eps = self._draw_noise(indices.shape[0])
        infos = self.enqueue_update(indices, eps).cpu().numpy()
        stats = self._mpo_stats.cpu().numpy()
        parallel.check_one_shot()
        replay.last_steps = steps
        for row, actor_row in zip(infos[0], stats):                 # mpo.py:92-97
            logger.store('critic/loss', row[0])
            logger.store('critic/q', row[1])
            for key, value in self.actor_updater.infos(actor_row).items():
                logger.store('actor/' + key, value)
        self.last_infos, self.last_actor_infos = infos, stats
        if self.model.observation_normalizer:
            self.model.observation_normalizer.update()


class TD3(DDPG):
    """tonic/torch/agents/td3.py:20-55."""

    policy_kind = 0

    def __init__(self, model=None, replay=None, exploration=None, actor_updater=None,
                 critic_updater=None, delay_steps=2):
        super().__init__(
            model=model or _twin_model(models.DeterministicPolicyHead()), replay=replay,
            exploration=exploration,
            actor_updater=actor_updater or updaters.DeterministicPolicyGradient(),
            critic_updater=critic_updater or updaters.TwinCriticDeterministicQLearning())
        self.delay_steps = delay_steps
        self.model.critic = self.model.critic_1        # td3.py:36 (also a checkpoint alias)

    def _actor_due(self, iteration):
        return (iteration + 1) % self.delay_steps == 0     # td3.py:43-46 (quirk Q8)

    def _draw_noise(self, iterations):
        # TargetActionNoise: one torch.randn_like(actions) per critic update (critics.py:131)
        B, A = self.replay.batch_size, self.action_size
        return np.stack([torch.randn(B, A).numpy()[None] for _ in range(iterations)])


class SAC(DDPG):
    """tonic/torch/agents/sac.py:22-51."""

    policy_kind = 1

    def __init__(self, model=None, replay=None, exploration=None, actor_updater=None,
                 critic_updater=None):
        head = models.GaussianPolicyHead(loc_activation=torch.nn.Identity,
                                         distribution=models.SquashedMultivariateNormalDiag)
        super().__init__(
            model=model or _twin_model(head), replay=replay,
            exploration=exploration or explorations.NoActionNoise(),
            actor_updater=actor_updater or updaters.TwinCriticSoftDeterministicPolicyGradient(),
            critic_updater=critic_updater or updaters.TwinCriticSoftQLearning())

    def _policy(self, observations):
        return self._forward_policy(observations, 1, True)        # sac.py:40-46

    def _draw_noise(self, iterations):
        # per iteration: rsample of the critic step, then rsample of the actor step
        B, A = self.replay.batch_size, self.action_size
        return np.stack([np.stack([torch.randn(B, A).numpy(), torch.randn(B, A).numpy()])
                         for _ in range(iterations)])
