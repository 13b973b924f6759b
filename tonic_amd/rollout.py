"""Device-resident synthetic rollout: the collect half of the hot path with every input
already in HBM (the benchmark contract's "inputs resident in HBM when the timed region
starts").  Observations, transition outcomes and the action noise of a whole Segment
(T time steps x W workers) are pre-generated on the device; ``collect`` then issues, per
environment step, exactly what ``PPO.step`` + ``PPO.update`` issue — policy forward + sample, Segment store and normaliser record — as ONE fused
``tonic_ppo_collect_step`` launch per environment step (``fused=False``: the separate
``tonic_ppo_act`` + ``tonic_segment_store`` launches the host-in-the-loop agent uses), without
the host in the loop.  The per-step launches are kept (one act per time step on W observations) because in
a real environment step t+1's observations depend on step t's actions.

``capture=True`` records the T-step sequence once into a hipGraph (via
``torch.cuda.CUDAGraph``; the C-ABI calls only enqueue kernels on the current stream, so they
are capturable) and replays it, removing the host launch cost.
"""
import torch

from tonic_amd import _lib


class DeviceRollout:
    def __init__(self, agent, workers, steps, seed=0, reset_probability=1e-3, fused=True,
                 packed=True):
        self.fused = fused
        self.packed = packed and fused
        self.agent = agent
        self.lib = _lib.load()
        device = agent.device
        O, A = agent.observation_size, agent.action_size
        updater = agent.actor_updater
        if getattr(updater, 'stock', False) or getattr(updater, 'torso', None) is not None or O > 32 or A > 8:
            raise NotImplementedError('DeviceRollout drives the fused per-step kernels: the default torso, '
                                      'O <= 32, A <= 8 (other agents collect through agent.step)')
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.W, self.T = workers, steps
        self.observations = torch.randn(steps + 1, workers, O, device=device, generator=gen)
        self.eps = torch.randn(steps, workers, A, device=device, generator=gen)
        self.rewards = torch.randn(steps, workers, device=device, generator=gen)
        resets = torch.rand(steps, workers, device=device, generator=gen) < reset_probability
        terms = resets & (torch.rand(steps, workers, device=device, generator=gen) < 0.5)
        self.resets, self.terminations = resets.float(), terms.float()
        self.actions = torch.empty(workers, A, device=device)
        self.log_probs = torch.empty(workers, device=device)
        self.graph = None

    def _enqueue(self):
        agent, lib, p = self.agent, self.lib, _lib.ptr
        replay = agent.replay
        if replay.buffers is None:
            replay._allocate(self.W, agent.observation_size, agent.action_size)
        b = replay.buffers
        norm = agent.model.observation_normalizer
        sums = norm.device_sums if norm is not None else None
        stream = _lib.current_stream()
        actor = p(agent.model.flat_actor.flat)
        O, A = agent.observation_size, agent.action_size
        if self.packed:
            if getattr(self, 'packed_actor', None) is None:
                self.packed_actor = torch.empty(lib.tonic_ppo_packed_actor_floats(O, A),
                                                device=agent.device)
            # once per collect (the parameters change at every learner update)
            _lib.check(lib.tonic_ppo_pack_actor(actor, p(self.packed_actor), O, A, stream),
                       'tonic_ppo_pack_actor')
        if self.packed:
            # one C call enqueues the T per-step launches (inputs are step-major and contiguous)
            _lib.check(lib.tonic_ppo_collect_steps_packed(
                p(self.packed_actor), p(self.observations), p(self.eps), p(self.rewards),
                p(self.resets), p(self.terminations), p(b['observations']), p(b['actions']),
                p(b['next_observations']), p(b['rewards']), p(b['resets']), p(b['terminations']),
                p(b['log_probs']), p(sums), 0, self.T, self.W, O, A, stream),
                'tonic_ppo_collect_steps_packed')
            return
        for t in range(self.T):
            if self.fused:
                _lib.check(lib.tonic_ppo_collect_step(
                    actor, p(self.observations[t]), p(self.eps[t]), p(self.observations[t + 1]),
                    p(self.rewards[t]), p(self.resets[t]), p(self.terminations[t]),
                    p(b['observations']), p(b['actions']), p(b['next_observations']),
                    p(b['rewards']), p(b['resets']), p(b['terminations']), p(b['log_probs']),
                    p(sums), None, t, self.W, O, A, stream), 'tonic_ppo_collect_step')
                continue
            _lib.check(lib.tonic_ppo_act(
                actor, p(self.observations[t]), p(self.eps[t]), p(self.actions),
                p(self.log_probs), self.W, O, A, stream), 'tonic_ppo_act')
            _lib.check(lib.tonic_segment_store(
                p(b['observations']), p(b['actions']), p(b['next_observations']),
                p(b['rewards']), p(b['resets']), p(b['terminations']), p(b['log_probs']),
                p(self.observations[t]), p(self.actions), p(self.observations[t + 1]),
                p(self.rewards[t]), p(self.resets[t]), p(self.terminations[t]),
                p(self.log_probs), p(sums), t, self.W, O, A, stream), 'tonic_segment_store')

    def collect(self, capture=False):
        """Fills the agent's Segment with T steps (asynchronous; no host sync)."""
        settle = getattr(self.agent, 'settle', None)
        if settle is not None:
            settle()        # (a PPO update may have left its critic iterations reading the Segment)
        if capture and not getattr(self, 'capture_failed', False):
            if self.graph is None:
                if self.packed and getattr(self, 'packed_actor', None) is None:
                    self.packed_actor = torch.empty(
                        self.lib.tonic_ppo_packed_actor_floats(self.agent.observation_size,
                                                               self.agent.action_size),
                        device=self.agent.device)
                # Warm-up outside the capture (buffer allocation, LDS opt-ins).  It runs the
                # rollout once, so undo its only cumulative side effect — the normaliser sums.
                norm = self.agent.model.observation_normalizer
                saved = norm.device_sums.clone() if norm is not None else None
                self._enqueue()
                if saved is not None:
                    norm.device_sums.copy_(saved)
                torch.cuda.synchronize()
                # thread_local: API calls of OTHER threads (e.g. the RCCL watchdog of a multi-GPU
                # run polling its events) must not invalidate this thread's capture.
                graph = torch.cuda.CUDAGraph()
                try:
                    with _lib.capturing(graph):
                        self._enqueue()
                    self.graph = graph
                except RuntimeError as error:       # still the HIP path, just launched eagerly
                    import sys
                    print(f'tonic_amd: hipGraph capture of the rollout failed ({error}); '
                          'falling back to eager kernel launches', file=sys.stderr)
                    self.capture_failed = True
                    torch.cuda.synchronize()
            if self.graph is not None:
                self.graph.replay()
            else:
                self._enqueue()
        else:
            self._enqueue()
        norm = self.agent.model.observation_normalizer
        if norm is not None:
            norm.new_count += self.T * self.W
        self.agent.replay.index = self.T
