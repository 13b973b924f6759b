"""Model containers for the MI355X engine, API-compatible with ``tonic.torch.models``.

Same public names, constructor arguments, two-phase ``__init__`` / ``initialize`` protocol
and ``state_dict`` key layout as the reference files
``tonic/torch/models/{utils,encoders,actors,critics,actor_critics}.py``, so checkpoints are
interchangeable with the reference and parameters are created by the same
``torch.nn.Linear`` default initialiser in the same order from the same CPU generator
(exact init parity).

What is different: after ``initialize`` every network (actor, critic, ...) is *packed* into
ONE contiguous float32 device buffer in ``parameters()`` order (``FlatNetwork``) and each
``nn.Parameter`` becomes a view into it.  The HIP kernels read / write those flat buffers
through the C ABI; the module tree only provides names, checkpointing and a stock-torch
``forward`` for code outside the accelerated path.
"""
import copy

import torch

FLOAT_EPSILON = 1e-8


class MLP(torch.nn.Module):
    """tonic/torch/models/utils.py:4-23."""

    def __init__(self, sizes, activation, fn=None):
        super().__init__()
        self.sizes, self.activation, self.fn = sizes, activation, fn

    def initialize(self, input_size):
        widths = [input_size, *self.sizes]
        modules = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            modules.extend((torch.nn.Linear(fan_in, fan_out), self.activation()))
        self.model = torch.nn.Sequential(*modules)
        if self.fn is not None:
            self.model.apply(self.fn)
        return widths[-1]

    def forward(self, inputs):
        return self.model(inputs)


def trainable_variables(model):
    return [p for p in model.parameters() if p.requires_grad]


class ObservationEncoder(torch.nn.Module):
    """tonic/torch/models/encoders.py:4-16."""

    def initialize(self, observation_space, action_space=None, observation_normalizer=None):
        self.observation_normalizer = observation_normalizer
        return observation_space.shape[0]

    def forward(self, observations):
        if self.observation_normalizer:
            observations = self.observation_normalizer(observations)
        return observations


class ObservationActionEncoder(torch.nn.Module):
    """tonic/torch/models/encoders.py:19-31."""

    def initialize(self, observation_space, action_space, observation_normalizer=None):
        self.observation_normalizer = observation_normalizer
        return observation_space.shape[0] + action_space.shape[0]

    def forward(self, observations, actions):
        if self.observation_normalizer:
            observations = self.observation_normalizer(observations)
        return torch.cat([observations, actions], dim=-1)


class SquashedMultivariateNormalDiag:
    """tonic/torch/models/actors.py:7-34."""

    def __init__(self, loc, scale):
        self._distribution = torch.distributions.normal.Normal(loc, scale)

    def rsample_with_log_prob(self, shape=()):
        raw = self._distribution.rsample(shape)
        squashed = torch.tanh(raw)
        log_probs = self._distribution.log_prob(raw) - torch.log(1 - squashed ** 2 + 1e-6)
        return squashed, log_probs

    def rsample(self, shape=()):
        return torch.tanh(self._distribution.rsample(shape))

    def sample(self, shape=()):
        return torch.tanh(self._distribution.sample(shape))

    def log_prob(self, samples):
        raise NotImplementedError('Use rsample_with_log_prob: unsquashed samples cannot be '
                                  'recovered accurately.')

    @property
    def loc(self):
        return torch.tanh(self._distribution.mean)


def _head_layer(input_size, output_size, activation, fn):
    layer = torch.nn.Sequential(torch.nn.Linear(input_size, output_size), activation())
    if fn:
        layer.apply(fn)
    return layer


class DetachedScaleGaussianPolicyHead(torch.nn.Module):
    """tonic/torch/models/actors.py:37-66 (state-independent log_scale parameter)."""

    def __init__(self, loc_activation=torch.nn.Tanh, loc_fn=None, log_scale_init=0.,
                 scale_min=1e-4, scale_max=1., distribution=torch.distributions.normal.Normal):
        super().__init__()
        self.loc_activation, self.loc_fn = loc_activation, loc_fn
        self.log_scale_init = log_scale_init
        self.scale_min, self.scale_max = scale_min, scale_max
        self.distribution = distribution

    def initialize(self, input_size, action_size):
        self.loc_layer = _head_layer(input_size, action_size, self.loc_activation, self.loc_fn)
        self.log_scale = torch.nn.Parameter(
            torch.full((1, action_size), self.log_scale_init, dtype=torch.float32))

    def forward(self, inputs):
        loc = self.loc_layer(inputs)
        scale = torch.nn.functional.softplus(self.log_scale) + FLOAT_EPSILON
        scale = torch.clamp(scale, self.scale_min, self.scale_max).repeat(inputs.shape[0], 1)
        return self.distribution(loc, scale)


class GaussianPolicyHead(torch.nn.Module):
    """tonic/torch/models/actors.py:69-98 (state-dependent scale; SAC)."""

    def __init__(self, loc_activation=torch.nn.Tanh, loc_fn=None,
                 scale_activation=torch.nn.Softplus, scale_min=1e-4, scale_max=1,
                 scale_fn=None, distribution=torch.distributions.normal.Normal):
        super().__init__()
        self.loc_activation, self.loc_fn = loc_activation, loc_fn
        self.scale_activation, self.scale_fn = scale_activation, scale_fn
        self.scale_min, self.scale_max = scale_min, scale_max
        self.distribution = distribution

    def initialize(self, input_size, action_size):
        self.loc_layer = _head_layer(input_size, action_size, self.loc_activation, self.loc_fn)
        self.scale_layer = _head_layer(input_size, action_size, self.scale_activation,
                                       self.scale_fn)

    def forward(self, inputs):
        scale = torch.clamp(self.scale_layer(inputs), self.scale_min, self.scale_max)
        return self.distribution(self.loc_layer(inputs), scale)


class DeterministicPolicyHead(torch.nn.Module):
    """tonic/torch/models/actors.py:101-115 (TD3 / DDPG)."""

    def __init__(self, activation=torch.nn.Tanh, fn=None):
        super().__init__()
        self.activation, self.fn = activation, fn

    def initialize(self, input_size, action_size):
        self.action_layer = _head_layer(input_size, action_size, self.activation, self.fn)

    def forward(self, inputs):
        return self.action_layer(inputs)


class ValueHead(torch.nn.Module):
    """tonic/torch/models/critics.py:4-20."""

    def __init__(self, fn=None):
        super().__init__()
        self.fn = fn

    def initialize(self, input_size, return_normalizer=None):
        self.return_normalizer = return_normalizer
        self.v_layer = torch.nn.Linear(input_size, 1)
        if self.fn:
            self.v_layer.apply(self.fn)

    def forward(self, inputs):
        out = self.v_layer(inputs).squeeze(-1)
        if self.return_normalizer:
            out = self.return_normalizer(out)
        return out


class CategoricalWithSupport:
    """tonic/torch/models/critics.py:23-46: a categorical distribution over the fixed support
    `values` (drop-in surface for callers of `model.critic(...)`; the learner kernels work on the
    logits, csrc/offpolicy.hip distributional_critic_loss_kernel)."""

    def __init__(self, values, logits):
        self.values = values
        self.logits = logits
        self.probabilities = torch.nn.functional.softmax(logits, dim=-1)

    def mean(self):
        return (self.probabilities * self.values).sum(dim=-1)

    def project(self, returns):
        """Probability mass of every atom of `returns` [B, NA] spread onto its two neighbours in
        the support (critics.py:32-46), as a triangular kernel per support point."""
        z = self.values
        vmin, vmax = z[0], z[-1]
        above = (torch.cat([z[1:], vmin[None]]) - z)[None, :, None]       # distance to the next atom
        below = (z - torch.cat([vmax[None], z[:-1]]))[None, :, None]      # ... to the previous one
        delta = torch.clamp(returns, vmin, vmax)[:, None] - z[None, :, None]
        up = (delta >= 0).float()
        hat = (up * delta / above) - ((1 - up) * delta / below)
        return (torch.clamp(1 - hat, 0, 1) * self.probabilities[:, None]).sum(dim=2)


class DistributionalValueHead(torch.nn.Module):
    """tonic/torch/models/critics.py:49-66."""

    def __init__(self, vmin, vmax, num_atoms, fn=None):
        super().__init__()
        self.num_atoms = num_atoms
        self.fn = fn
        self.values = torch.linspace(vmin, vmax, num_atoms).float()

    def initialize(self, input_size, return_normalizer=None):
        if return_normalizer:
            raise ValueError('Return normalizers cannot be used with distributional value heads.')
        self.distributional_layer = torch.nn.Linear(input_size, self.num_atoms)
        if self.fn:
            self.distributional_layer.apply(self.fn)

    def forward(self, inputs):
        logits = self.distributional_layer(inputs)
        return CategoricalWithSupport(values=self.values.to(logits.device), logits=logits)


class _Network(torch.nn.Module):
    """encoder -> torso -> head (actors.py:118-137, critics.py:70-90)."""

    def __init__(self, encoder, torso, head):
        super().__init__()
        self.encoder, self.torso, self.head = encoder, torso, head

    def forward(self, *inputs):
        return self.head(self.torso(self.encoder(*inputs)))


class Actor(_Network):
    def initialize(self, observation_space, action_space, observation_normalizer=None):
        # Quirk Q1: the reference passes the normaliser POSITIONALLY into
        # the encoder's `action_space` slot (actors.py:128-129 vs encoders.py:5-8), so torch
        # actors never normalise observations.  Reproduced on purpose.
        size = self.encoder.initialize(observation_space, observation_normalizer)
        size = self.torso.initialize(size)
        self.head.initialize(size, action_space.shape[0])


class Critic(_Network):
    def initialize(self, observation_space, action_space, observation_normalizer=None,
                   return_normalizer=None):
        size = self.encoder.initialize(
            observation_space=observation_space, action_space=action_space,
            observation_normalizer=observation_normalizer)
        size = self.torso.initialize(size)
        self.head.initialize(size, return_normalizer)


def network_variables(module):
    """Weights / biases of a network in ``parameters()`` order, WITHOUT the shared normaliser
    statistics (non-trainable ``_mean`` / ``_std``) — target networks included even though
    their ``requires_grad`` is False."""
    return [p for name, p in module.named_parameters() if 'normalizer' not in name]


class FlatNetwork:
    """One or several networks' parameters packed into a single contiguous device buffer, in
    ``parameters()`` order (the layout the C ABI documents); every ``nn.Parameter`` becomes a
    view of it, so ``state_dict`` / ``load_state_dict`` keep working and the kernels see one
    flat pointer.  ``out`` packs into an existing (zeroed) buffer region instead of allocating.

    ``padded`` selects the off-policy parameter layout of include/tonic_hip.h: every tensor on a
    16-byte boundary, weight rows ``tonic_mlp_weight_stride(cols)`` floats apart (the weight
    parameters are then strided views; the padding stays zero)."""

    @staticmethod
    def slots(modules, padded):
        """[(parameter, floats in the buffer, row stride or None)] in buffer order."""
        if isinstance(modules, torch.nn.Module):
            modules = [modules]
        params = [p for module in modules for p in network_variables(module)]
        if not padded:
            return [(p, p.numel(), None) for p in params]
        from tonic_amd import _lib
        stride = _lib.load().tonic_mlp_weight_stride
        out = []
        for p in params:
            if p.dim() == 2:
                ld = int(stride(p.shape[1]))
                out.append((p, p.shape[0] * ld, ld))
            else:
                out.append((p, (p.numel() + 3) // 4 * 4, None))
        return out

    @staticmethod
    def length(modules, padded=False):
        return sum(n for _, n, _ in FlatNetwork.slots(modules, padded))

    def __init__(self, modules, device, out=None, padded=False):
        slots = self.slots(modules, padded)
        self.count = sum(n for _, n, _ in slots)
        self.flat = out if out is not None else torch.zeros(
            self.count, dtype=torch.float32, device=device)
        assert self.flat.numel() == self.count
        offset = 0
        for p, n, ld in slots:
            if ld is None:
                view = self.flat[offset:offset + p.numel()].view(p.shape)
            else:
                view = self.flat[offset:offset + n].view(p.shape[0], ld)[:, :p.shape[1]]
            view.copy_(p.data)
            p.data = view
            offset += n
        self.params = [p for p, _, _ in slots]

    def shapes(self):
        return [tuple(p.shape) for p in self.params]


class ActorCritic(torch.nn.Module):
    """tonic/torch/models/actor_critics.py:8-29."""

    def __init__(self, actor, critic, observation_normalizer=None, return_normalizer=None):
        super().__init__()
        self.actor, self.critic = actor, critic
        self.observation_normalizer = observation_normalizer
        self.return_normalizer = return_normalizer

    def initialize(self, observation_space, action_space):
        if self.observation_normalizer:
            self.observation_normalizer.initialize(observation_space.shape)
        self.actor.initialize(observation_space, action_space, self.observation_normalizer)
        self.critic.initialize(observation_space, action_space, self.observation_normalizer,
                               self.return_normalizer)

    def pack(self, device):
        """Moves the model to `device` and packs actor / critic into flat buffers."""
        self.to(device)
        self.flat_actor = FlatNetwork(self.actor, device)
        self.flat_critic = FlatNetwork(self.critic, device)
        return self


class ActorCriticWithTargets(torch.nn.Module):
    """tonic/torch/models/actor_critics.py:32-72 (DDPG family)."""

    def __init__(self, actor, critic, observation_normalizer=None, return_normalizer=None,
                 target_coeff=0.005):
        super().__init__()
        self.actor, self.critic = actor, critic
        self.target_actor, self.target_critic = copy.deepcopy(actor), copy.deepcopy(critic)
        self.observation_normalizer = observation_normalizer
        self.return_normalizer = return_normalizer
        self.target_coeff = target_coeff

    def _networks(self):
        return [(self.actor, False), (self.critic, True), (self.target_actor, False),
                (self.target_critic, True)]

    def initialize(self, observation_space, action_space):
        if self.observation_normalizer:
            self.observation_normalizer.initialize(observation_space.shape)
        for net, is_critic in self._networks():
            extra = (self.return_normalizer,) if is_critic else ()
            net.initialize(observation_space, action_space, self.observation_normalizer, *extra)
        online = [n for n, _ in self._networks()[:len(self._networks()) // 2]]
        target = [n for n, _ in self._networks()[len(self._networks()) // 2:]]
        self.online_variables = [p for n in online for p in trainable_variables(n)]
        self.target_variables = [p for n in target for p in trainable_variables(n)]
        for p in self.target_variables:
            p.requires_grad = False
        self.assign_targets()

    def assign_targets(self):
        for o, t in zip(self.online_variables, self.target_variables):
            t.data.copy_(o.data)

    def pack(self, device):
        """Moves the model to `device`; the online networks share one flat buffer
        [actor | critic(s)] (off-policy parameter layout: padded weight rows) and the targets
        another with the same layout, so ``update_targets`` is one polyak launch and all
        critics are one Adam block."""
        self.to(device)
        nets = [n for n, _ in self._networks()]
        online, target = nets[:len(nets) // 2], nets[len(nets) // 2:]
        total = FlatNetwork.length(online, padded=True)
        self.flat_online = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_target = torch.zeros(total, dtype=torch.float32, device=device)
        n_actor = FlatNetwork.length(online[0], padded=True)
        self.flat_actor = FlatNetwork(online[0], device, self.flat_online[:n_actor], True)
        self.flat_critics = FlatNetwork(online[1:], device, self.flat_online[n_actor:], True)
        self.flat_target_actor = FlatNetwork(target[0], device, self.flat_target[:n_actor], True)
        self.flat_target_critics = FlatNetwork(target[1:], device, self.flat_target[n_actor:], True)
        return self

    def update_targets(self):
        """actor_critics.py:68-72 / 126-130 as ONE launch over the flat online / target buffers
        (the per-parameter host loop only before ``pack``)."""
        if getattr(self, 'flat_online', None) is None:
            with torch.no_grad():
                for o, t in zip(self.online_variables, self.target_variables):
                    t.data.mul_(1 - self.target_coeff)
                    t.data.add_(self.target_coeff * o.data)
            return
        from tonic_amd import _lib
        _lib.check(_lib.load().tonic_polyak_update(
            _lib.ptr(self.flat_target), _lib.ptr(self.flat_online), self.flat_online.numel(),
            float(self.target_coeff), _lib.current_stream()), 'tonic_polyak_update')


class ActorTwinCriticWithTargets(ActorCriticWithTargets):
    """tonic/torch/models/actor_critics.py:75-130 (TD3 / SAC)."""

    def __init__(self, actor, critic, observation_normalizer=None, return_normalizer=None,
                 target_coeff=0.005):
        torch.nn.Module.__init__(self)
        self.actor = actor
        self.critic_1, self.critic_2 = critic, copy.deepcopy(critic)
        self.target_actor = copy.deepcopy(actor)
        self.target_critic_1, self.target_critic_2 = copy.deepcopy(critic), copy.deepcopy(critic)
        self.observation_normalizer = observation_normalizer
        self.return_normalizer = return_normalizer
        self.target_coeff = target_coeff

    def _networks(self):
        return [(self.actor, False), (self.critic_1, True), (self.critic_2, True),
                (self.target_actor, False), (self.target_critic_1, True),
                (self.target_critic_2, True)]

