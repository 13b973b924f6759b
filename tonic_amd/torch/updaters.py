"""Learner updaters on the HIP engine — API of ``tonic/torch/updaters/{actors,critics}.py``.

``ClippedRatio`` (actors.py:53-112) and ``VRegression`` (critics.py:6-28) keep their
constructor arguments, ``initialize(model)`` and ``__call__`` returning a dict whose values
expose ``.numpy()``.  Internally one update is three asynchronous launches on the current
stream, with no host synchronisation:

    tonic_ppo_actor_grad / tonic_value_regression_grad   fused forward + loss + backward
    [all-reduce of the flat gradient-sum buffer over RCCL when world_size > 1]
    tonic_adam_step                                       flat Adam + logged statistics

The PPO agent drives them through ``enqueue`` for all 80 iterations and reads the
statistics back once; ``__call__`` (enqueue + read back) exists for drop-in compatibility.
"""
import os

import numpy as np
import torch

from tonic_amd import _lib

INFO_WIDTH = 8
ACTOR_INFO = ('loss', 'kl', 'entropy', 'clip_fraction', 'std', 'stop')


def adam_hyperparameters(factory, default_lr):
    """Reads lr / betas / eps from the reference-style ``optimizer=lambda params: Adam(...)``
    factory (actors.py:58-59) so user configs keep working; only plain Adam is fused."""
    if factory is None:
        return dict(lr=default_lr, betas=(0.9, 0.999), eps=1e-8)
    probe = factory([torch.nn.Parameter(torch.zeros(1))])
    if not isinstance(probe, torch.optim.Adam) or isinstance(probe, torch.optim.AdamW):
        raise NotImplementedError(f'only torch.optim.Adam is fused on the HIP engine, got {type(probe)}')
    d = probe.defaults
    if d.get('weight_decay', 0) != 0 or d.get('amsgrad', False) or d.get('maximize', False):
        raise NotImplementedError('Adam with weight_decay / amsgrad / maximize is not fused')
    return dict(lr=float(d['lr']), betas=tuple(float(b) for b in d['betas']), eps=float(d['eps']))


def fused_ppo_torso(torso):
    """The fused PPO kernels (csrc/mlp64x16.hip; layer by layer for wide observations / actions) serve the
    reference's default torso, MLP((64, 64), Tanh)."""
    return tuple(torso.sizes) == (64, 64) and torso.activation is torch.nn.Tanh


_TORSO_ACTIVATIONS = {torch.nn.Tanh: 1, torch.nn.ReLU: 2}


def hip_ppo_torso(torso):
    """(layers, sizes as a ctypes int32 array, activation code) when the layer-by-layer HIP path
    (csrc/mlpwide.hip, the tonic_*_torso entries) serves this ``MLP(sizes, activation)``: 1 .. 4 hidden
    layers of 4 .. 256 units (multiples of 4), Tanh or ReLU, no custom initialiser — None otherwise (and for
    the default torso, which the fused kernels serve).  Every other torso runs the same arithmetic as stock
    PyTorch-ROCm operators on the HBM-resident data (see `_StockTorch`).  The entries themselves take layers
    up to 384 units (tested), but beyond 256 a layer is several 64-output slices each re-reading its input
    rows and rocBLAS wins — (384, 300) ReLU at N = 1 M: 41.4 ms vs 32.3 ms per learner iteration,
    profiles/r05_torso_timing.txt — so the agents hand such torsos to the stock operators
    (TONIC_AMD_TORSO_WIDE_HIP=1 keeps them on the HIP entries)."""
    import ctypes
    sizes = tuple(int(v) for v in torso.sizes)
    code = _TORSO_ACTIVATIONS.get(torso.activation)
    if fused_ppo_torso(torso) or code is None or not 1 <= len(sizes) <= 4:
        return None
    widest = 384 if os.environ.get('TONIC_AMD_TORSO_WIDE_HIP', '0') == '1' else 256
    if any(v < 4 or v > widest or v % 4 for v in sizes):
        return None
    if os.environ.get('TONIC_AMD_TORSO_STOCK', '0') == '1':      # (developer switch: A/B against stock torch)
        return None
    return len(sizes), (ctypes.c_int32 * len(sizes))(*sizes), code


class _StockTorch:
    """Networks outside the shapes of the hand-written kernels (tonic/torch/models/utils.py:4-23
    accepts any ``MLP(sizes, activation)``): forward, loss, autograd backward and the optimizer
    step are stock PyTorch-ROCm operators on the device-resident Segment / Buffer batches and on
    the very parameter views of the flat buffers — same data layout, same agent loops, same
    statistics rows as the fused path; still no CPU path (the agents refuse to initialise without
    a GPU).  Slower than the fused kernels and with one host read-back per step for the flags the
    fused path keeps on the device; pinned on goldens of the reference's agents with such torsos
    (tests/golden/ppo_relu256_small.npz, sac_uneven_small.npz)."""

    stock = False

    def _stock_setup(self, factory, default_lr):
        self.stock = True
        variables = [p for p in self.variables]
        for p in variables:
            p.requires_grad_(True)
        self.torch_optimizer = (factory(variables) if factory is not None
                                else torch.optim.Adam(variables, lr=default_lr))

    def _stock_reduce(self):
        """Several ranks (equal shards): the per-rank mean gradients are averaged."""
        from tonic_amd import parallel
        if not parallel.exchanging():
            return
        world = parallel.world_size()
        for p in self.variables:
            if p.grad is not None:
                torch.distributed.all_reduce(p.grad)
                p.grad /= world

    def _stock_apply(self):
        self._stock_reduce()
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_(self.variables, self.gradient_clip)
        self.torch_optimizer.step()
        self.steps_enqueued += 1
        self.state[0] += 1


class _FlatUpdater(_StockTorch):
    stats_kind = 0
    torso = None            # (layers, sizes, activation): the tonic_*_torso entries serve this network

    def _setup(self, flat, hyper):
        self.lib = _lib.load()
        self.flat = flat
        self.hyper = hyper
        device = flat.flat.device
        self.count = flat.count
        self.grad_sums = torch.zeros(self.count + INFO_WIDTH, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.state = torch.zeros(4, dtype=torch.int32, device=device)   # {step, stop, -, -}
        self.steps_enqueued = 0     # host mirror of state[0] (off-policy: every enqueued step is taken)
        self.workspace = None
        self.scratch_info = torch.zeros(INFO_WIDTH, dtype=torch.float32, device=device)
        self.gradient_clip = float(getattr(self, 'gradient_clip', 0) or 0)
        self.clip_workspace = torch.zeros(self.lib.tonic_clip_workspace_bytes(self.count),
                                          dtype=torch.uint8, device=device)
        # Workgroups of the fused grad launches (0 = the kernel's own width).  Part of the
        # updater's configuration, not of a call: the grouping of the float32 partial sums follows
        # it, so results are reproducible as long as it stays what it is (agents.PPO sets the
        # critic's once, from the worker count — whether or not its iterations then run under
        # the next rollout).
        self.max_workgroups = 0
        self.world_size = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world_size = torch.distributed.get_world_size()

    def _workspace_for(self, n):
        """Scratch of the grad kernels: per-workgroup partial images for the fused kernels; for
        shapes beyond them (O > 32, A > 8: csrc/mlpwide.hip) also the activations of the
        layer-by-layer passes."""
        actor = self.stats_kind == 1
        if self.torso is not None:
            need = self.lib.tonic_ppo_torso_workspace_bytes(
                n, self.observation_size, self.action_size if actor else 1, 1 if actor else 0,
                self.torso[0], self.torso[1])
        else:
            need = self.lib.tonic_ppo_workspace_bytes(
                n, self.observation_size, self.action_size if actor else 1, 1 if actor else 0)
        if need < 0:
            raise NotImplementedError(
                f'PPO networks with {self.observation_size} observations / '
                f'{getattr(self, "action_size", 1)} actions are outside the HIP kernels '
                '(O <= 384, A <= 32)')
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.grad_sums.device)
        return self.workspace

    def share_gradient_buffer(self, buffer):
        """Makes `grad_sums` a view of a caller-owned buffer so that several updaters' gradient
        sums travel in ONE all-reduce (the caller then passes allreduce=False to enqueue)."""
        assert buffer.numel() == self.count + INFO_WIDTH
        self.grad_sums = buffer

    def enqueue_clip(self, n_global, skip=None):
        """clip_grad_norm_ between backward and the optimizer step (actors.py:96-98,
        critics.py:24-25): on the (all-reduced) gradient sums, in place."""
        if self.gradient_clip > 0:
            _lib.check(self.lib.tonic_clip_grad_norm(
                _lib.ptr(self.grad_sums), self.count, 1.0 / n_global, self.gradient_clip, skip,
                _lib.ptr(self.clip_workspace), self.clip_workspace.numel(),
                _lib.current_stream()), 'tonic_clip_grad_norm')

    def _step(self, n_global, info_row, adv_stats=None, skip=None, kl_threshold=0.0,
              entropy_coeff=0.0, allreduce=True, targets=None):
        """All-reduce of the gradient sums (world > 1) + Adam + statistics.  `targets` =
        (flat target buffer, flat online buffer, offset of this block in them, coeff): the
        polyak update of ALL targets rides in the same launch (update_targets right after the
        step, as ddpg.py:105-112 orders them)."""
        from tonic_amd import parallel
        self.steps_enqueued += 1
        if parallel.exchanging() and allreduce:
            one_shot = parallel.one_shot(self.count + INFO_WIDTH)
            if one_shot is not None:
                one_shot.all_reduce(self.grad_sums)              # tonic_allreduce_f32
            else:
                torch.distributed.all_reduce(self.grad_sums)     # RCCL sum over xGMI
        self.enqueue_clip(n_global, skip)
        h = self.hyper
        if targets is not None:
            target, online, offset, coeff = targets
            assert online.data_ptr() + 4 * offset == self.flat.flat.data_ptr()
            _lib.check(self.lib.tonic_adam_polyak_step(
                _lib.ptr(online), _lib.ptr(self.grad_sums), _lib.ptr(self.exp_avg),
                _lib.ptr(self.exp_avg_sq), _lib.ptr(self.state), offset, self.count,
                online.numel(), 1.0 / n_global, h['lr'], h['betas'][0], h['betas'][1], h['eps'],
                self.stats_kind, _lib.ptr(info_row), _lib.ptr(target), float(coeff),
                _lib.current_stream()), 'tonic_adam_polyak_step')
            return
        _lib.check(self.lib.tonic_adam_step(
            _lib.ptr(self.flat.flat), _lib.ptr(self.grad_sums), _lib.ptr(self.exp_avg),
            _lib.ptr(self.exp_avg_sq), _lib.ptr(self.state), self.count, 1.0 / n_global,
            h['lr'], h['betas'][0], h['betas'][1], h['eps'], self.stats_kind,
            float(kl_threshold), float(entropy_coeff), _lib.ptr(adv_stats),
            _lib.ptr(info_row), skip, _lib.current_stream()), 'tonic_adam_step')


def adam_step_constants(hyper, step):
    """{step_size, bias_correction2_sqrt} of optimizer step `step` (1-based) as float32 of the
    float64 values torch.optim.Adam forms in Python (adam.py:530-536)."""
    beta1, beta2 = hyper['betas']
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    return hyper['lr'] / bias_correction1, bias_correction2 ** 0.5


def enqueue_step_pair(actor, critic, n_local, adv_stats, actor_info, critic_info):
    """The optimizer steps of a PPO iteration (ppo.py:33-46: actor, then critic) as ONE launch
    pair.  The gradient sums are final (all-reduced by the caller when world > 1)."""
    ha, hc = actor.hyper, critic.hyper
    if actor.stock or critic.stock or (ha['betas'], ha['eps']) != (hc['betas'], hc['eps']):   # separately
        actor.enqueue_step(n_local, adv_stats, actor_info, allreduce=False)
        critic.enqueue_step(n_local, critic_info, allreduce=False)
        return
    n_global = n_local * actor.world_size
    actor.enqueue_clip(n_global, actor.stop_flag_ptr())
    critic.enqueue_clip(n_global)
    p = _lib.ptr
    _lib.check(actor.lib.tonic_adam_step_pair(
        p(actor.flat.flat), p(actor.grad_sums), p(actor.exp_avg), p(actor.exp_avg_sq),
        p(actor.state), actor.count, ha['lr'], actor.stats_kind, float(actor.kl_threshold),
        float(actor.entropy_coeff), p(adv_stats), p(actor_info), actor.stop_flag_ptr(),
        p(critic.flat.flat), p(critic.grad_sums), p(critic.exp_avg), p(critic.exp_avg_sq),
        p(critic.state), critic.count, hc['lr'], critic.stats_kind, p(critic_info),
        1.0 / (n_local * actor.world_size), ha['betas'][0], ha['betas'][1], ha['eps'],
        _lib.current_stream()), 'tonic_adam_step_pair')


class ClippedRatio(_FlatUpdater):
    stats_kind = 1

    def __init__(self, optimizer=None, ratio_clip=0.2, kl_threshold=0.015, entropy_coeff=0,
                 gradient_clip=0):
        self.optimizer = optimizer
        self.ratio_clip = ratio_clip
        self.kl_threshold = kl_threshold
        self.entropy_coeff = entropy_coeff
        self.gradient_clip = gradient_clip

    def initialize(self, model):
        self.model = model
        self._setup(model.flat_actor, adam_hyperparameters(self.optimizer, 3e-4))
        self.variables = model.flat_actor.params
        self.observation_size = model.actor.torso.model[0].in_features
        self.action_size = model.actor.head.log_scale.shape[1]
        if not fused_ppo_torso(model.actor.torso):
            self.torso = hip_ppo_torso(model.actor.torso)
            if self.torso is None or self.lib.tonic_ppo_torso_param_count(
                    self.observation_size, self.action_size, 1, self.torso[0], self.torso[1]) != self.count:
                self.torso = None
                self._stock_setup(self.optimizer, 3e-4)

    def _stock_grad(self, observations, actions, advantages, adv_stats, log_probs):
        """actors.py:70-112 up to the optimizer step (ClippedRatio; ratio_clip < 0: the plain
        policy gradient of actors.py:22-51).  `advantages` are raw: normalised here with the
        segment's statistics like `get_full` does (segments.py:41-46)."""
        self._stock_row = None
        if int(self.state[1]) != 0:                      # stopped earlier in this update
            return
        mean, std, all_zero, normalise = (float(v) for v in adv_stats.tolist())
        if normalise:
            advantages = (advantages - mean) / std
        zero = torch.zeros((), device=observations.device)
        if all_zero:                                      # actors.py:71-75 / 23-26
            with torch.no_grad():
                distributions = self.model.actor(observations)
                self._stock_row = (zero, zero, distributions.entropy().mean(), zero,
                                   distributions.stddev.mean(), False)
            return
        self.torch_optimizer.zero_grad()
        distributions = self.model.actor(observations)
        new_log_probs = distributions.log_prob(actions).sum(dim=-1)
        entropy = distributions.entropy().mean()
        if self.ratio_clip >= 0:
            ratios_1 = torch.exp(new_log_probs - log_probs)
            surrogates_1 = advantages * ratios_1
            ratios_2 = torch.clamp(ratios_1, 1 - self.ratio_clip, 1 + self.ratio_clip)
            loss = -torch.min(surrogates_1, advantages * ratios_2).mean()
        else:
            loss = -(advantages * new_log_probs).mean()
        if self.entropy_coeff != 0:
            loss = loss - self.entropy_coeff * entropy
        loss.backward()
        with torch.no_grad():
            kl = (log_probs - new_log_probs).mean()
            if self.ratio_clip >= 0:
                clipped = ratios_1.gt(1 + self.ratio_clip) | ratios_1.lt(1 - self.ratio_clip)
                clip_fraction = clipped.float().mean()
            else:
                clip_fraction = zero
            self._stock_row = (loss.detach(), kl, entropy.detach(), clip_fraction,
                               distributions.stddev.mean().detach(), True)

    def _stock_step(self, info_row):
        if self._stock_row is None:
            return
        loss, kl, entropy, clip_fraction, std, step = self._stock_row
        if step:
            self._stock_apply()
        stop = bool(kl > self.kl_threshold)               # actors.py:112
        info_row[:5] = torch.stack([loss, kl, entropy, clip_fraction, std])
        info_row[5] = float(stop)
        info_row[6] = 1.0
        if stop:
            self.state[1] = 1

    def stop_flag_ptr(self):
        return self.state.data_ptr() + 4

    def reset_stop(self):
        self.state[1:2].zero_()

    def enqueue_grad(self, observations, actions, advantages, adv_stats, log_probs):
        if self.stock:
            return self._stock_grad(observations, actions, advantages, adv_stats, log_probs)
        n = observations.shape[0]
        ws = self._workspace_for(n)
        p = _lib.ptr
        if self.torso is not None:
            _lib.check(self.lib.tonic_ppo_actor_grad_torso(
                *self.torso, p(self.flat.flat), p(observations), p(actions), p(advantages), p(adv_stats),
                p(log_probs), p(self.grad_sums), n, self.observation_size, self.action_size,
                float(self.ratio_clip), float(self.entropy_coeff), self.stop_flag_ptr(),
                p(ws), ws.numel(), _lib.current_stream()), 'tonic_ppo_actor_grad_torso')
            return
        _lib.check(self.lib.tonic_ppo_actor_grad(
            p(self.flat.flat), p(observations), p(actions), p(advantages), p(adv_stats),
            p(log_probs), p(self.grad_sums), n, self.observation_size, self.action_size,
            float(self.ratio_clip), float(self.entropy_coeff), self.stop_flag_ptr(),
            self.max_workgroups, p(ws), ws.numel(), _lib.current_stream()), 'tonic_ppo_actor_grad')

    def enqueue_step(self, n_local, adv_stats, info_row, allreduce=True):
        if self.stock:
            return self._stock_step(info_row)
        self._step(n_local * self.world_size, info_row, adv_stats, self.stop_flag_ptr(),
                   self.kl_threshold, self.entropy_coeff, allreduce)

    def enqueue(self, observations, actions, advantages, adv_stats, log_probs, info_row):
        self.enqueue_grad(observations, actions, advantages, adv_stats, log_probs)
        self.enqueue_step(observations.shape[0], adv_stats, info_row)

    def __call__(self, observations, actions, advantages, log_probs):
        """Drop-in form (actors.py:70-112): `advantages` are final (already normalised)."""
        all_zero = bool((advantages == 0).all())
        adv_stats = torch.tensor([0., 1., float(all_zero), 0.], device=advantages.device)
        self.reset_stop()
        self.scratch_info.zero_()
        self.enqueue(observations, actions, advantages, adv_stats, log_probs, self.scratch_info)
        row = self.scratch_info.cpu()
        self.reset_stop()
        out = {k: row[i].clone() for i, k in enumerate(ACTOR_INFO)}
        out['stop'] = out['stop'] > 0.5
        return out


class StochasticPolicyGradient(ClippedRatio):
    """actors.py:9-51 (A2C): loss = -mean(advantages * log_probs) [- entropy_coeff * entropy], one
    Adam step.  Same fused kernel as ClippedRatio in its plain mode (`ratio_clip < 0` in the C ABI:
    the gradient of -adv * logp is -adv, nothing to clip) and no KL stop."""

    def __init__(self, optimizer=None, entropy_coeff=0, gradient_clip=0):
        super().__init__(optimizer=optimizer, ratio_clip=-1.0, kl_threshold=float('inf'),
                         entropy_coeff=entropy_coeff, gradient_clip=gradient_clip)

    def __call__(self, observations, actions, advantages, log_probs):
        out = super().__call__(observations, actions, advantages, log_probs)
        return {k: out[k] for k in ('loss', 'kl', 'entropy', 'std')}


class ConjugateGradient:
    """optimizers.py:25-115 — truncated natural gradient with a backtracking line search, on the
    device: the search direction solves H x = g by `conjugate_gradient_steps` CG iterations whose
    Hessian-vector products are autograd double-backward passes of the constraint (KL) over the
    HBM-resident batch; then x is scaled to the trust region and shrunk until the constraint holds
    and the loss did not rise.  The CG scalars stay 0-dim device tensors (no host sync inside the
    loop); each backtracking trial reads two scalars back, like the reference.

    Several ranks (equal shards of the batch): the loss / constraint values, the gradient and every
    Hessian-vector product are batch MEANS, so each is averaged over the ranks with one all-reduce
    and all ranks walk through identical CG iterations and backtracking trials."""

    def __init__(self, conjugate_gradient_steps=10, damping_coefficient=0.1,
                 constraint_threshold=0.01, backtrack_steps=10, backtrack_coefficient=0.8):
        self.conjugate_gradient_steps = conjugate_gradient_steps
        self.damping_coefficient = damping_coefficient
        self.constraint_threshold = constraint_threshold
        self.backtrack_steps = backtrack_steps
        self.backtrack_coefficient = backtrack_coefficient

    def optimize(self, loss_function, constraint_function, variables):
        from tonic_amd import parallel
        eps = 1e-8                                              # optimizers.py:5
        world = parallel.world_size() if parallel.exchanging() else 1

        def across(tensor):
            """Mean over the ranks of a per-rank batch mean (in place; one rank: nothing)."""
            if world > 1:
                torch.distributed.all_reduce(tensor)
                tensor /= world
            return tensor

        def flat(tensors):
            return torch.cat([t.reshape(-1) for t in tensors])

        def hessian_vector(x):                                  # optimizers.py:36-48
            first = flat(torch.autograd.grad(constraint_function(), variables, create_graph=True))
            second = across(flat(torch.autograd.grad((first * x).sum(), variables)))
            if self.damping_coefficient > 0:
                second = second + self.damping_coefficient * x
            return second

        def assign(values):                                     # optimizers.py:13-22
            offset = 0
            with torch.no_grad():
                for v in variables:
                    v.copy_(values[offset:offset + v.numel()].view(v.shape))
                    offset += v.numel()

        def trial(scale):                                       # optimizers.py:68-74
            assign(start - alpha * direction * scale)
            with torch.no_grad():
                both = across(torch.stack([constraint_function(), loss_function()]))
                return both[0], both[1]

        start = flat([v.detach() for v in variables]).clone()
        loss = loss_function()
        gradient = across(flat(torch.autograd.grad(loss, variables)))
        start_loss = float(across(loss.detach().clone()))
        zero = torch.zeros((), dtype=torch.float32)
        residual_dot = gradient.dot(gradient)
        if float(residual_dot) == 0:                            # optimizers.py:55-56, 87-91
            return zero, zero, torch.as_tensor(0, dtype=torch.int32)
        direction = torch.zeros_like(gradient)
        residual, search = gradient.clone(), gradient.clone()
        for _ in range(self.conjugate_gradient_steps):          # optimizers.py:58-65
            z = hessian_vector(search)
            step = residual_dot / (search.dot(z) + eps)
            direction += step * search
            residual -= step * z
            new_dot = residual.dot(residual)
            search = residual + (new_dot / residual_dot) * search
            residual_dot = new_dot
        alpha = torch.sqrt(2 * self.constraint_threshold /
                           direction.dot(hessian_vector(direction)) + eps)      # optimizers.py:93-94
        if self.backtrack_steps is None or self.backtrack_coefficient is None:
            constraint, loss = trial(1)
            return constraint.cpu(), loss.cpu()
        for i in range(self.backtrack_steps):                   # optimizers.py:101-113
            constraint, loss = trial(self.backtrack_coefficient ** i)
            if float(constraint) <= self.constraint_threshold and float(loss) <= start_loss:
                break
            if i == self.backtrack_steps - 1:
                constraint, loss = trial(0)
                i = self.backtrack_steps
        return constraint.cpu(), loss.cpu(), torch.as_tensor(i + 1, dtype=torch.int32)


class TrustRegionPolicyGradient:
    """actors.py:115-156 (TRPO).  The one updater of this package that is not a fused HIP kernel:
    SURVEY.md §8(f4) scopes A2C / TRPO as the stock-torch path — the loss, the KL and the
    Fisher-vector products are PyTorch autograd over `model.actor`, whose parameters ARE views of
    the flat HBM buffer the act kernels read, on the HBM-resident Segment (no host copy of the
    batch).  `locs` / `scales` of the behaviour policy may be omitted: the parameters have not
    moved since the rollout, so the actor itself reproduces them."""

    def __init__(self, optimizer=None, entropy_coeff=0):
        self.optimizer = optimizer or ConjugateGradient()
        self.entropy_coeff = entropy_coeff

    def initialize(self, model):
        self.model = model
        self.variables = list(model.flat_actor.params)
        self.world_size = 1

    def __call__(self, observations, actions, log_probs, advantages, locs=None, scales=None):
        from tonic_amd import parallel
        device = self.variables[0].device
        observations, actions, log_probs, advantages = (
            torch.as_tensor(v, dtype=torch.float32, device=device)
            for v in (observations, actions, log_probs, advantages))
        if locs is None or scales is None:
            with torch.no_grad():
                behaviour = self.model.actor(observations)
                locs, scales = behaviour.loc, behaviour.stddev
        else:
            locs, scales = (torch.as_tensor(v, dtype=torch.float32, device=device)
                            for v in (locs, scales))
        nothing = (advantages == 0.).all().float()             # actors.py:127-130 (on ALL ranks)
        if parallel.exchanging():
            torch.distributed.all_reduce(nothing, op=torch.distributed.ReduceOp.MIN)
        if bool(nothing):
            zero = torch.zeros((), dtype=torch.float32)
            return dict(loss=zero, kl=zero, backtrack_steps=torch.as_tensor(0, dtype=torch.int32))
        kl, loss, steps = self.optimizer.optimize(
            loss_function=lambda: self._loss(observations, actions, log_probs, advantages),
            constraint_function=lambda: self._kl(observations, locs, scales),
            variables=self.variables)
        return dict(loss=loss, kl=kl, backtrack_steps=steps)

    def _loss(self, observations, actions, old_log_probs, advantages):        # actors.py:142-150
        distributions = self.model.actor(observations)
        log_probs = distributions.log_prob(actions).sum(dim=-1)
        loss = -(torch.exp(log_probs - old_log_probs) * advantages).mean()
        if self.entropy_coeff != 0:
            loss = loss - self.entropy_coeff * distributions.entropy().mean()
        return loss

    def _kl(self, observations, locs, scales):                                # actors.py:152-156
        distributions = self.model.actor(observations)
        behaviour = type(distributions)(locs, scales)
        return torch.distributions.kl.kl_divergence(distributions, behaviour).mean()


class VRegression(_FlatUpdater):
    stats_kind = 2

    def __init__(self, loss=None, optimizer=None, gradient_clip=0):
        if loss is not None and not isinstance(loss, torch.nn.MSELoss):
            raise NotImplementedError('only the default MSE loss is fused')
        self.loss = loss
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip

    def initialize(self, model):
        self.model = model
        self._setup(model.flat_critic, adam_hyperparameters(self.optimizer, 1e-3))
        self.variables = model.flat_critic.params
        self.observation_size = model.critic.torso.model[0].in_features
        self.normalizer = model.observation_normalizer
        if not fused_ppo_torso(model.critic.torso):
            self.torso = hip_ppo_torso(model.critic.torso)
            if self.torso is None or self.lib.tonic_ppo_torso_param_count(
                    self.observation_size, 1, 0, self.torso[0], self.torso[1]) != self.count:
                self.torso = None
                self._stock_setup(self.optimizer, 1e-3)
        if self.normalizer is None:
            device = self.grad_sums.device
            self._unit_mean = torch.zeros(self.observation_size, device=device)
            self._unit_std = torch.ones(self.observation_size, device=device)

    def norm_tensors(self):
        if self.normalizer is None:
            return self._unit_mean, self._unit_std
        return self.normalizer._mean.data, self.normalizer._std.data

    def norm_clip(self):
        """MeanStd(clip=...) (mean_stds.py:37-38) as the C ABI wants it: 0 = None."""
        return float(getattr(self.normalizer, 'clip', None) or 0.0)

    def forward_values(self, observations, out):
        if self.stock:
            with torch.no_grad():
                out.copy_(self.model.critic(observations))
            return out
        mean, std = self.norm_tensors()
        p = _lib.ptr
        ws = self._workspace_for(observations.shape[0])
        if self.torso is not None:
            _lib.check(self.lib.tonic_value_forward_torso(
                *self.torso, p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(observations), p(out),
                observations.shape[0], self.observation_size, p(ws), ws.numel(),
                _lib.current_stream()), 'tonic_value_forward_torso')
            return out
        _lib.check(self.lib.tonic_value_forward_wide(
            p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(observations), p(out),
            observations.shape[0], self.observation_size, p(ws), ws.numel(),
            _lib.current_stream()), 'tonic_value_forward_wide')
        return out

    def enqueue_grad(self, observations, returns, norm=None):
        """`norm`: (mean, std) to use instead of the normaliser's tensors — a snapshot, for
        iterations that run while the normaliser is being updated (agents.PPO._update)."""
        if self.stock:                                    # critics.py:18-28
            self.torch_optimizer.zero_grad()
            values = self.model.critic(observations)
            loss = torch.nn.functional.mse_loss(values, returns)
            loss.backward()
            self._stock_row = (loss.detach(), values.detach().mean())
            return
        n = observations.shape[0]
        ws = self._workspace_for(n)
        mean, std = norm if norm is not None else self.norm_tensors()
        p = _lib.ptr
        if self.torso is not None:
            _lib.check(self.lib.tonic_value_regression_grad_torso(
                *self.torso, p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(observations),
                p(returns), p(self.grad_sums), n, self.observation_size, p(ws), ws.numel(),
                _lib.current_stream()), 'tonic_value_regression_grad_torso')
            return
        _lib.check(self.lib.tonic_value_regression_grad(
            p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(observations), p(returns),
            p(self.grad_sums), n, self.observation_size, self.max_workgroups, p(ws), ws.numel(),
            _lib.current_stream()),
            'tonic_value_regression_grad')

    def enqueue_step(self, n_local, info_row, allreduce=True):
        if self.stock:
            self._stock_apply()
            info_row[:2] = torch.stack(self._stock_row)
            info_row[6] = 1.0
            return
        self._step(n_local * self.world_size, info_row, allreduce=allreduce)

    def enqueue(self, observations, returns, info_row):
        self.enqueue_grad(observations, returns)
        self.enqueue_step(observations.shape[0], info_row)

    def __call__(self, observations, returns):
        """Drop-in form (critics.py:18-28).  `v` is the pre-step value vector."""
        values = torch.empty(observations.shape[0], device=observations.device)
        self.forward_values(observations, values)
        self.scratch_info.zero_()
        self.enqueue(observations, returns, self.scratch_info)
        row = self.scratch_info.cpu()
        return dict(loss=row[0].clone(), v=values.cpu())


# ----------------------------------------------------------------- off-policy (SAC / TD3)

_Q_ACTIVATIONS = {torch.nn.ReLU: 1, torch.nn.Tanh: 2, torch.nn.ELU: 3}       # GemmAct of csrc/gemm16.h


def _torso_width(torso, generic=False):
    """The `H` argument of the off-policy entries: the width of the reference's torso — two ReLU layers of one
    width, which the fused kernels hold (csrc/mlpfwd.hip) — or, `generic`, tonic_mlp_hidden(H1, H2, activation)
    for any other two-layer torso with ReLU / Tanh / ELU (unequal widths: the (400, 300) class), which the SAC /
    TD3 / DDPG entries run layer by layer on gemm16 launches.  Anything else raises (callers with a stock-torch
    form catch it)."""
    sizes = tuple(int(v) for v in torso.sizes)
    plain = len(sizes) == 2 and sizes[0] == sizes[1] and torso.activation is torch.nn.ReLU
    if plain:                              # (a plain width goes through the C ABI's `H` as is)
        return sizes[0]
    code = _Q_ACTIVATIONS.get(torso.activation)
    if (generic and len(sizes) == 2 and code is not None
            and os.environ.get('TONIC_AMD_TORSO_STOCK', '0') != '1'):
        packed = _lib.load().tonic_mlp_hidden(sizes[0], sizes[1], code)
        if packed > 0:
            return packed
    raise NotImplementedError('the off-policy kernels serve two-layer torsos (ReLU / Tanh / ELU for SAC, TD3 '
                              f'and DDPG; equal-width ReLU for D4PG and MPO), got {sizes} / {torso.activation}')


class _QUpdater(_FlatUpdater):
    """Shared plumbing of the four off-policy updaters: shapes, normaliser tensors, workspace."""

    # updaters whose stock-torch form exists (`_stock_enqueue`): the others keep raising for torsos
    # outside the fused kernels
    stock_capable = False

    def _shapes(self, model):
        self.model = model
        self.observation_size = model.actor.torso.model[0].in_features
        critic = model.critic_1 if hasattr(model, 'critic_1') else model.critic
        try:
            self.hidden = _torso_width(model.actor.torso, self.stock_capable)
            if _torso_width(critic.torso, self.stock_capable) != self.hidden:
                raise NotImplementedError('actor and critic torsos must have the same shape')
        except NotImplementedError:
            if not self.stock_capable:
                raise
            self.hidden = None           # any MLP(sizes, activation): stock torch operators
        head = model.actor.head
        self.sac = hasattr(head, 'scale_layer')
        layer = head.loc_layer if self.sac else head.action_layer
        self.action_size = layer[0].out_features
        self.normalizer = model.observation_normalizer
        device = model.flat_online.device
        self.atoms = getattr(critic.head, 'num_atoms', 0)
        if self.normalizer is None:
            self._unit = (torch.zeros(self.observation_size, device=device),
                          torch.ones(self.observation_size, device=device))
        if self.hidden is None:
            return
        # the C side walks the flat blocks by its own offsets: both must agree on the layout
        heads = 2 if self.sac else 1
        lib = _lib.load()
        want_actor = lib.tonic_mlp_actor_param_count(
            self.observation_size, self.hidden, self.action_size, heads)
        self.atoms = getattr(critic.head, 'num_atoms', 0)          # > 0: distributional critic (D4PG)
        if self.atoms:
            if not 2 <= self.atoms <= 64:
                raise NotImplementedError('the distributional kernels serve 2 .. 64 atoms')
            want_critic = lib.tonic_mlp_actor_param_count(
                self.observation_size + self.action_size, self.hidden, self.atoms, 1)
            self.values = critic.head.values.to(device=device, dtype=torch.float32).contiguous()
            # the critic step takes returns and projection from the TARGET critic's distribution
            # (critics.py:104-109); its support is the online one unless somebody changed one head
            # after the model was built (the targets are copies made at construction)
            target = getattr(model, 'target_critic', critic)
            if getattr(target.head, 'num_atoms', self.atoms) != self.atoms:
                raise NotImplementedError('online and target critic with different numbers of atoms')
            self.target_values = target.head.values.to(device=device,
                                                       dtype=torch.float32).contiguous()
        else:
            want_critic = lib.tonic_q_critic_param_count(
                self.observation_size, self.action_size, self.hidden)
        n_critics = 2 if hasattr(model, 'critic_1') else 1
        if (model.flat_actor.count, model.flat_critics.count) != (want_actor,
                                                                   n_critics * want_critic):
            raise NotImplementedError(
                f'network shapes outside the packed off-policy layout: actor '
                f'{model.flat_actor.count} vs {want_actor}, critics {model.flat_critics.count} '
                f'vs {n_critics} x {want_critic}')
        if self.normalizer is None:
            self._unit = (torch.zeros(self.observation_size, device=device),
                          torch.ones(self.observation_size, device=device))

    def norm_tensors(self):
        if self.normalizer is None:
            return self._unit
        return self.normalizer._mean.data, self.normalizer._std.data

    def norm_clip(self):
        return float(getattr(self.normalizer, 'clip', None) or 0.0)

    def _offpolicy_workspace(self, batch):
        if self.atoms:
            need = self.lib.tonic_distributional_workspace_bytes(
                batch, self.observation_size, self.action_size, self.hidden, self.atoms)
        else:
            need = self.lib.tonic_offpolicy_workspace_bytes(batch, self.observation_size,
                                                            self.action_size, self.hidden)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.grad_sums.device)
        return self.workspace

    def enqueue_empty(self, info_row, n_global, targets=None):
        """This rank drew none of the global batch: contribute zero sums, take the same step."""
        self.grad_sums.zero_()
        self._step(n_global, info_row, targets=targets)

    def _info(self, fn, keys):
        self.scratch_info.zero_()
        fn(self.scratch_info)
        row = self.scratch_info.cpu()
        return {k: row[i].clone() for i, k in enumerate(keys)}


class TargetActionNoise:
    """critics.py:125-134 (parameters only: the clipping runs inside tonic_twin_q_grad)."""

    def __init__(self, scale=0.2, clip=0.5):
        self.scale, self.clip = scale, clip


def squashed_sample(distribution, eps):
    """SquashedMultivariateNormalDiag.rsample_with_log_prob (models/actors.py:11-16) with the
    standard-normal draws given: tanh(loc + scale * eps) and its summed log-probability."""
    normal = distribution._distribution
    raw = normal.mean + normal.stddev * eps
    squashed = torch.tanh(raw)
    log_probs = normal.log_prob(raw) - torch.log(1 - squashed ** 2 + 1e-6)
    return squashed, log_probs.sum(dim=-1)


class _TwinCriticQLearning(_QUpdater):
    stock_capable = True
    stats_kind = 3          # {loss, q1 mean, q2 mean}
    default_lr = 1e-3
    kind = 0

    def initialize(self, model):
        self._shapes(model)
        self._setup(model.flat_critics, adam_hyperparameters(self.optimizer, self.default_lr))
        self.variables = model.flat_critics.params
        if self.hidden is None:
            self._stock_setup(self.optimizer, self.default_lr)

    def _policy_params(self):
        raise NotImplementedError

    def _stock_enqueue(self, batch, eps, info_row, n_global):
        """critics.py:68-86 (kind 2), :156-182 (kind 0), :202-235 (kind 1) as stock torch operators;
        `eps` are the standard-normal draws the reference makes inside (target-action noise /
        the next action's rsample), in its order.  Losses are formed as SUMS over this rank's part
        of the batch divided by the global batch size, so that several ranks add up to the mean."""
        from tonic_amd import parallel
        model, B = self.model, batch['observations'].shape[0]
        n = float(n_global or B)
        twin = hasattr(model, 'critic_1')
        with torch.no_grad():
            if self.kind == 1:
                actions, log_probs = squashed_sample(model.actor(batch['next_observations']), eps)
            else:
                actions = model.target_actor(batch['next_observations'])
                if self.kind == 0:
                    noise = self.target_action_noise
                    actions = actions + torch.clamp(noise.scale * eps, -noise.clip, noise.clip)
                    actions = torch.clamp(actions, -1, 1)
            if twin:
                next_values = torch.min(
                    model.target_critic_1(batch['next_observations'], actions),
                    model.target_critic_2(batch['next_observations'], actions))
            else:
                next_values = model.target_critic(batch['next_observations'], actions)
            if self.kind == 1:
                next_values = next_values - self.entropy_coeff * log_probs
            returns = batch['rewards'] + batch['discounts'] * next_values
        self.torch_optimizer.zero_grad()
        critics = (model.critic_1, model.critic_2) if twin else (model.critic,)
        values = [critic(batch['observations'], batch['actions']) for critic in critics]
        loss = sum(((v - returns) ** 2).sum() for v in values) / n
        loss.backward()
        row = torch.stack([loss.detach()] + [v.detach().sum() / n for v in values])
        if parallel.exchanging():
            torch.distributed.all_reduce(row)
            for p in self.variables:
                torch.distributed.all_reduce(p.grad)
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_(self.variables, self.gradient_clip)
        self.torch_optimizer.step()
        self.steps_enqueued += 1
        self.state[0] += 1
        info_row[:row.numel()] = row
        info_row[6] = 1.0

    def enqueue(self, batch, eps, info_row, n_global=None):
        if self.stock:
            return self._stock_enqueue(batch, eps, info_row, n_global)
        B = batch['observations'].shape[0]
        ws = self._offpolicy_workspace(B)
        mean, std = self.norm_tensors()
        noise = getattr(self, 'target_action_noise', None)
        p = _lib.ptr
        _lib.check(self.lib.tonic_twin_q_grad(
            self.kind, p(self._policy_params()), p(self.model.flat_target_critics.flat),
            p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(batch['observations']),
            p(batch['actions']),
            p(batch['next_observations']), p(batch['rewards']), p(batch['discounts']), p(eps),
            p(self.grad_sums), B, self.observation_size, self.hidden, self.action_size,
            float(getattr(self, 'entropy_coeff', 0.0)), float(noise.scale if noise else 0.0),
            float(noise.clip if noise else 0.0), p(ws), ws.numel(), _lib.current_stream()),
            'tonic_twin_q_grad')
        self._step(n_global or B * self.world_size, info_row)

    def __call__(self, observations, actions, next_observations, rewards, discounts):
        """Drop-in form: draws its own noise from the torch CPU generator like the reference."""
        batch = dict(observations=observations, actions=actions,
                     next_observations=next_observations, rewards=rewards, discounts=discounts)
        eps = torch.randn(actions.shape).to(actions.device)
        out = self._info(lambda row: self.enqueue(batch, eps, row), ('loss', 'q1', 'q2'))
        return out


class DeterministicQLearning(_TwinCriticQLearning):
    """critics.py:56-86 (DDPG): ONE critic, target actor, no target noise."""
    kind, default_lr = 2, 1e-3

    def __init__(self, loss=None, optimizer=None, gradient_clip=0):
        _check_plain(loss, gradient_clip)
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip

    def _policy_params(self):
        return self.model.flat_target_actor.flat

    def __call__(self, observations, actions, next_observations, rewards, discounts):
        batch = dict(observations=observations, actions=actions,
                     next_observations=next_observations, rewards=rewards, discounts=discounts)
        out = self._info(lambda row: self.enqueue(batch, None, row), ('loss', 'q'))
        return out


class TwinCriticDeterministicQLearning(_TwinCriticQLearning):
    """critics.py:137-182 (TD3): target actor + clipped target-action noise."""
    kind, default_lr = 0, 1e-3

    def __init__(self, loss=None, optimizer=None, target_action_noise=None, gradient_clip=0):
        _check_plain(loss, gradient_clip)
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip
        self.target_action_noise = target_action_noise or TargetActionNoise(scale=0.2, clip=0.5)

    def _policy_params(self):
        return self.model.flat_target_actor.flat


class TwinCriticSoftQLearning(_TwinCriticQLearning):
    """critics.py:185-235 (SAC): online-actor sample and entropy bonus in the target."""
    kind, default_lr = 1, 3e-4

    def __init__(self, loss=None, optimizer=None, entropy_coeff=0.2, gradient_clip=0):
        _check_plain(loss, gradient_clip)
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip
        self.entropy_coeff = entropy_coeff

    def _policy_params(self):
        return self.model.flat_actor.flat


class DistributionalDeterministicQLearning(_TwinCriticQLearning):
    """critics.py:89-122 (D4PG): cross-entropy of the online critic's categorical distribution
    against the projected target distribution (tonic_distributional_q_grad)."""
    stock_capable = False
    stats_kind = 4          # {loss}

    def __init__(self, optimizer=None, gradient_clip=0):
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip

    def initialize(self, model):
        super().initialize(model)
        if not self.atoms:
            raise NotImplementedError('DistributionalDeterministicQLearning needs a critic with a '
                                      'DistributionalValueHead')

    def enqueue(self, batch, eps, info_row, n_global=None):
        B = batch['observations'].shape[0]
        ws = self._offpolicy_workspace(B)
        mean, std = self.norm_tensors()
        p = _lib.ptr
        _lib.check(self.lib.tonic_distributional_q_grad(
            p(self.model.flat_target_actor.flat), p(self.model.flat_target_critics.flat),
            p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(batch['observations']),
            p(batch['actions']), p(batch['next_observations']), p(batch['rewards']),
            p(batch['discounts']), p(self.target_values), p(self.grad_sums), B, self.observation_size,
            self.hidden, self.action_size, self.atoms, p(ws), ws.numel(), _lib.current_stream()),
            'tonic_distributional_q_grad')
        self._step(n_global or B * self.world_size, info_row)

    def __call__(self, observations, actions, next_observations, rewards, discounts):
        batch = dict(observations=observations, actions=actions,
                     next_observations=next_observations, rewards=rewards, discounts=discounts)
        return self._info(lambda row: self.enqueue(batch, None, row), ('loss',))


class ExpectedSARSA(_TwinCriticQLearning):
    """critics.py:238-282 (MPO): one critic regressed on r + discount * the mean target value of
    `num_samples` actions drawn from the target actor (tonic_expected_sarsa_grad)."""
    default_lr = 3e-4
    stock_capable = False

    def __init__(self, num_samples=20, loss=None, optimizer=None, gradient_clip=0):
        _check_plain(loss, gradient_clip)
        self.num_samples = num_samples
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip

    def _offpolicy_workspace(self, batch):
        need = self.lib.tonic_mpo_workspace_bytes(batch, self.observation_size, self.action_size,
                                                  self.hidden, self.num_samples)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.grad_sums.device)
        return self.workspace

    def enqueue(self, batch, eps, info_row, n_global=None):
        """eps: the [num_samples * B, A] standard-normal draws of `rsample((num_samples,))`."""
        B = batch['observations'].shape[0]
        ws = self._offpolicy_workspace(B)
        mean, std = self.norm_tensors()
        p = _lib.ptr
        _lib.check(self.lib.tonic_expected_sarsa_grad(
            p(self.model.flat_target_actor.flat), p(self.model.flat_target_critics.flat),
            p(self.flat.flat), p(mean), p(std), self.norm_clip(), p(batch['observations']),
            p(batch['actions']), p(batch['next_observations']), p(batch['rewards']),
            p(batch['discounts']), p(eps), p(self.grad_sums), B, self.observation_size, self.hidden,
            self.action_size, self.num_samples, p(ws), ws.numel(), _lib.current_stream()),
            'tonic_expected_sarsa_grad')
        self._step(n_global or B, info_row)

    def __call__(self, observations, actions, next_observations, rewards, discounts):
        batch = dict(observations=observations, actions=actions,
                     next_observations=next_observations, rewards=rewards, discounts=discounts)
        eps = torch.randn(self.num_samples * observations.shape[0], self.action_size).to(
            observations.device)
        return self._info(lambda row: self.enqueue(batch, eps, row), ('loss', 'q'))


class _ActorQGradient(_QUpdater):
    stock_capable = True
    stats_kind = 4          # {loss}
    default_lr = 1e-3
    kind = 0

    def initialize(self, model):
        self._shapes(model)
        self._setup(model.flat_actor, adam_hyperparameters(self.optimizer, self.default_lr))
        self.variables = model.flat_actor.params
        if self.hidden is None:
            self._stock_setup(self.optimizer, self.default_lr)

    def _stock_enqueue(self, observations, eps, info_row, n_global, targets):
        """actors.py:170-189 (kind 0: -mean q of `model.critic`) / :238-267 (kind 1: mean of
        alpha log-prob - min q) as stock torch operators; only the actor's variables receive
        gradients (the reference freezes the critics around its backward)."""
        from tonic_amd import parallel
        model, B = self.model, observations.shape[0]
        n = float(n_global or B)
        if self.kind == 1:
            actions, log_probs = squashed_sample(model.actor(observations), eps)
            values = torch.min(model.critic_1(observations, actions),
                               model.critic_2(observations, actions))
            loss = (self.entropy_coeff * log_probs - values).sum() / n
        else:
            critic = model.critic if hasattr(model, 'critic') else model.critic_1
            loss = -critic(observations, model.actor(observations)).sum() / n
        grads = torch.autograd.grad(loss, self.variables)
        row = loss.detach().reshape(1)
        for p, grad in zip(self.variables, grads):
            p.grad = grad
        if parallel.exchanging():
            torch.distributed.all_reduce(row)
            for p in self.variables:
                torch.distributed.all_reduce(p.grad)
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_(self.variables, self.gradient_clip)
        self.torch_optimizer.step()
        self.steps_enqueued += 1
        self.state[0] += 1
        info_row[0] = row[0]
        info_row[6] = 1.0
        if targets is not None:                         # update_targets right after (ddpg.py:112)
            model.update_targets()

    def enqueue(self, observations, eps, info_row, n_global=None, targets=None):
        if self.stock:
            return self._stock_enqueue(observations, eps, info_row, n_global, targets)
        B = observations.shape[0]
        ws = self._offpolicy_workspace(B)
        mean, std = self.norm_tensors()
        p = _lib.ptr
        _lib.check(self.lib.tonic_actor_q_grad(
            self.kind, p(self.flat.flat), p(self.model.flat_critics.flat), p(mean), p(std),
            self.norm_clip(), p(observations), p(eps), p(self.grad_sums), B, self.observation_size, self.hidden,
            self.action_size, float(getattr(self, 'entropy_coeff', 0.0)), p(ws), ws.numel(),
            _lib.current_stream()), 'tonic_actor_q_grad')
        self._step(n_global or B * self.world_size, info_row, targets=targets)

    def __call__(self, observations):
        eps = None
        if self.kind == 1:
            eps = torch.randn(observations.shape[0], self.action_size).to(observations.device)
        return self._info(lambda row: self.enqueue(observations, eps, row), ('loss',))


class DeterministicPolicyGradient(_ActorQGradient):
    """actors.py:159-189 on `model.critic` (= critic_1 for TD3, td3.py:36)."""
    kind, default_lr = 0, 1e-3

    def __init__(self, optimizer=None, gradient_clip=0):
        _check_plain(None, gradient_clip)
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip


class DistributionalDeterministicPolicyGradient(_ActorQGradient):
    """actors.py:192-224 (D4PG): ascend the mean of the critic's value distribution
    (tonic_distributional_actor_grad)."""
    stock_capable = False

    def __init__(self, optimizer=None, gradient_clip=0):
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip

    def initialize(self, model):
        super().initialize(model)
        if not self.atoms:
            raise NotImplementedError('DistributionalDeterministicPolicyGradient needs a critic '
                                      'with a DistributionalValueHead')

    def enqueue(self, observations, eps, info_row, n_global=None, targets=None):
        B = observations.shape[0]
        ws = self._offpolicy_workspace(B)
        mean, std = self.norm_tensors()
        p = _lib.ptr
        _lib.check(self.lib.tonic_distributional_actor_grad(
            p(self.flat.flat), p(self.model.flat_critics.flat), p(mean), p(std), self.norm_clip(),
            p(observations), p(self.values), p(self.grad_sums), B, self.observation_size,
            self.hidden, self.action_size, self.atoms, p(ws), ws.numel(), _lib.current_stream()),
            'tonic_distributional_actor_grad')
        self._step(n_global or B * self.world_size, info_row, targets=targets)


MPO_INFO = ('policy_mean_loss', 'policy_std_loss', 'kl_mean_loss', 'kl_std_loss', 'alpha_mean_loss',
            'alpha_std_loss', 'temperature_loss', 'temperature')


class MaximumAPosterioriPolicyOptimization(_ActorQGradient):
    """actors.py:270-464 with per-dimension KL constraints (tonic_mpo_actor_grad): the actor step
    and the step of the dual variables {log_temperature, log_alpha_mean[A], log_alpha_std[A],
    log_penalty_temperature} — one flat device vector with its own Adam state (actors.py:289-316;
    like the reference, the dual optimizer is Adam(lr=1e-2) unless `actor_optimizer` is given)."""
    default_lr = 3e-4
    stock_capable = False

    def __init__(self, num_samples=20, epsilon=1e-1, epsilon_penalty=1e-3, epsilon_mean=1e-3,
                 epsilon_std=1e-6, initial_log_temperature=1., initial_log_alpha_mean=1.,
                 initial_log_alpha_std=10., min_log_dual=-18., per_dim_constraining=True,
                 action_penalization=True, actor_optimizer=None, dual_optimizer=None,
                 gradient_clip=0):
        if not per_dim_constraining:
            raise NotImplementedError('only per-dimension KL constraints are fused')
        _check_plain(None, gradient_clip)
        self.num_samples = num_samples
        self.epsilon, self.epsilon_penalty = epsilon, epsilon_penalty
        self.epsilon_mean, self.epsilon_std = epsilon_mean, epsilon_std
        self.initial_log_temperature = initial_log_temperature
        self.initial_log_alpha_mean = initial_log_alpha_mean
        self.initial_log_alpha_std = initial_log_alpha_std
        self.min_log_dual = min_log_dual
        self.action_penalization = action_penalization
        self.optimizer = actor_optimizer
        self.dual_hyper = adam_hyperparameters(actor_optimizer, 1e-2)      # actors.py:291-292
        self.gradient_clip = gradient_clip

    def initialize(self, model, action_space=None):
        super().initialize(model)
        A, device = self.action_size, self.grad_sums.device
        duals = [self.initial_log_temperature] + [self.initial_log_alpha_mean] * A + \
            [self.initial_log_alpha_std] * A + [self.initial_log_temperature]
        self.duals = torch.tensor(duals, dtype=torch.float32, device=device)
        self.dual_grads = torch.zeros(2 * A + 2 + INFO_WIDTH, dtype=torch.float32, device=device)
        self.dual_exp_avg = torch.zeros(2 * A + 2, dtype=torch.float32, device=device)
        self.dual_exp_avg_sq = torch.zeros(2 * A + 2, dtype=torch.float32, device=device)
        self.dual_state = torch.zeros(4, dtype=torch.int32, device=device)
        self.mpo_stats = torch.zeros(9 + 2 * A, dtype=torch.float32, device=device)
        self.dual_info = torch.zeros(INFO_WIDTH, dtype=torch.float32, device=device)
        self.column_sums = torch.zeros(6 + 2 * A, dtype=torch.float64, device=device)
        self.dual_clip_workspace = torch.zeros(
            self.lib.tonic_clip_workspace_bytes(2 * A + 2), dtype=torch.uint8, device=device)

    def _offpolicy_workspace(self, batch):
        need = self.lib.tonic_mpo_workspace_bytes(batch, self.observation_size, self.action_size,
                                                  self.hidden, self.num_samples)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.grad_sums.device)
        return self.workspace

    def enqueue(self, observations, eps, info_row, n_global=None, targets=None, stats_row=None):
        from tonic_amd import parallel
        p = _lib.ptr
        # (actors.py:347-356 floors the log-duals in place at the head of the call: the kernels read them
        #  through the floor and write the floored values back ahead of the duals' optimizer step)
        floor = float(self.min_log_dual)
        stats = stats_row if stats_row is not None else self.mpo_stats
        B = 0 if observations is None else observations.shape[0]
        if parallel.exchanging():
            # this rank's part of the global batch (possibly nothing): the actor's gradient sums and
            # the column sums of the per-state terms; the dual step needs their GLOBAL means
            if B > 0:
                ws = self._offpolicy_workspace(B)
                mean, std = self.norm_tensors()
                _lib.check(self.lib.tonic_mpo_actor_grad_shard(
                    p(self.flat.flat), p(self.model.flat_target_actor.flat),
                    p(self.model.flat_target_critics.flat), p(self.duals), floor, p(mean), p(std),
                    self.norm_clip(), p(observations), p(eps), p(self.grad_sums),
                    p(self.column_sums), B, self.observation_size, self.hidden, self.action_size,
                    self.num_samples, int(bool(self.action_penalization)), p(ws), ws.numel(),
                    _lib.current_stream()), 'tonic_mpo_actor_grad_shard')
            else:
                self.grad_sums.zero_()
                self.column_sums.zero_()
            torch.distributed.all_reduce(self.column_sums)
            _lib.check(self.lib.tonic_mpo_dual_step(
                p(self.column_sums), p(self.duals), floor, p(self.dual_grads), p(stats),
                p(self.grad_sums[self.count:]), B, n_global, self.action_size, self.num_samples,
                float(self.epsilon), float(self.epsilon_penalty), float(self.epsilon_mean),
                float(self.epsilon_std), int(bool(self.action_penalization)),
                _lib.current_stream()), 'tonic_mpo_dual_step')
        else:
            ws = self._offpolicy_workspace(B)
            mean, std = self.norm_tensors()
            _lib.check(self.lib.tonic_mpo_actor_grad(
                p(self.flat.flat), p(self.model.flat_target_actor.flat),
                p(self.model.flat_target_critics.flat), p(self.duals), floor, p(mean), p(std),
                self.norm_clip(), p(observations), p(eps), p(self.grad_sums), p(self.dual_grads),
                p(stats), B, self.observation_size, self.hidden, self.action_size,
                self.num_samples, float(self.epsilon), float(self.epsilon_penalty),
                float(self.epsilon_mean), float(self.epsilon_std),
                int(bool(self.action_penalization)), p(ws), ws.numel(), _lib.current_stream()),
                'tonic_mpo_actor_grad')
        self._step(n_global or B, info_row, targets=targets)
        h = self.dual_hyper
        if self.gradient_clip > 0:
            # actors.py:441-445 clips the dual variables' gradient norm as well (the penalty
            # temperature's entry is zero without action penalisation, so it does not count)
            _lib.check(self.lib.tonic_clip_grad_norm(
                p(self.dual_grads), self.duals.numel(), 1.0, self.gradient_clip, None,
                p(self.dual_clip_workspace), self.dual_clip_workspace.numel(),
                _lib.current_stream()), 'tonic_clip_grad_norm (duals)')
        _lib.check(self.lib.tonic_adam_step(
            p(self.duals), p(self.dual_grads), p(self.dual_exp_avg), p(self.dual_exp_avg_sq),
            p(self.dual_state), self.duals.numel(), 1.0, h['lr'], h['betas'][0], h['betas'][1],
            h['eps'], 0, 0.0, 0.0, None, p(self.dual_info), None, _lib.current_stream()),
            'tonic_adam_step (duals)')

    def enqueue_empty(self, info_row, n_global, targets=None, stats_row=None):
        """This rank drew none of the global batch: zero sums, the same dual and actor steps."""
        self.enqueue(None, None, info_row, n_global, targets, stats_row)

    def infos(self, stats):
        """The reference's return dict (actors.py:449-464) from one statistics row (host array)."""
        A = self.action_size
        out = {k: stats[i] for i, k in enumerate(MPO_INFO)}
        out['alpha_mean'] = stats[8:8 + A]
        out['alpha_std'] = stats[8 + A:8 + 2 * A]
        if self.action_penalization:
            out['penalty_temperature'] = stats[8 + 2 * A]
        return out

    def __call__(self, observations):
        eps = torch.randn(self.num_samples * observations.shape[0], self.action_size).to(
            observations.device)
        self.scratch_info.zero_()
        self.enqueue(observations, eps, self.scratch_info)
        return {k: torch.as_tensor(v) for k, v in self.infos(self.mpo_stats.cpu().numpy()).items()}


class TwinCriticSoftDeterministicPolicyGradient(_ActorQGradient):
    """actors.py:226-267 (SAC)."""
    kind, default_lr = 1, 3e-4

    def __init__(self, optimizer=None, entropy_coeff=0.2, gradient_clip=0):
        _check_plain(None, gradient_clip)
        self.optimizer = optimizer
        self.gradient_clip = gradient_clip
        self.entropy_coeff = entropy_coeff


def _check_plain(loss, gradient_clip):
    if loss is not None and not isinstance(loss, torch.nn.MSELoss):
        raise NotImplementedError('only the default MSE loss is fused')
