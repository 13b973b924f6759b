"""TEST / BASELINE INFRASTRUCTURE ONLY — the reference's PPO path restated on the SAME
torch-CPU operators the reference executes (``torch.nn.Linear``, ``torch.distributions.Normal``,
autograd, ``torch.optim.Adam``) so it can be (a) timed on the GPU box's host cores as
``cpu_baseline`` (kind "port": ``/root/reference`` does not exist there) and (b) used as a
full-size float32 checker.  Pinned against the golden vectors by tests/test_oracle_golden.py.

Cited reference code (relative to the reference checkout): ``tonic/torch/agents/a2c.py:41-99``,
``ppo.py:20-67``, ``tonic/torch/updaters/actors.py:70-112``, ``critics.py:18-28``,
``tonic/replays/segments.py:27-78``, ``tonic/torch/normalizers/mean_stds.py:34-74``; TRPO:
``tonic/torch/agents/trpo.py:7-97``, ``updaters/actors.py:115-156``, ``updaters/optimizers.py:25-115``.
"""
import numpy as np
import torch

import numpy_port as port


class TorchPPO:
    def __init__(self, observation_size, action_size, seed=0, steps=4096, iterations=80,
                 actor_lr=3e-4, critic_lr=1e-3):
        torch.manual_seed(seed)
        O, A = observation_size, action_size
        tanh = torch.nn.Tanh
        # creation order = reference init order (actor torso, head, critic torso, head)
        self.actor_torso = torch.nn.Sequential(torch.nn.Linear(O, 64), tanh(),
                                               torch.nn.Linear(64, 64), tanh())
        self.loc_layer = torch.nn.Sequential(torch.nn.Linear(64, A), tanh())
        self.log_scale = torch.nn.Parameter(torch.zeros(1, A))
        self.critic_torso = torch.nn.Sequential(torch.nn.Linear(O, 64), tanh(),
                                                torch.nn.Linear(64, 64), tanh())
        self.v_layer = torch.nn.Linear(64, 1)
        self.actor_vars = [*self.actor_torso.parameters(), self.log_scale,
                           *self.loc_layer.parameters()]
        self.critic_vars = [*self.critic_torso.parameters(), *self.v_layer.parameters()]
        self.actor_opt = torch.optim.Adam(self.actor_vars, lr=actor_lr)
        self.critic_opt = torch.optim.Adam(self.critic_vars, lr=critic_lr)
        self.normalizer = port.MeanStdPort((O,))
        self.norm_mean = torch.zeros(O)
        self.norm_std = torch.ones(O)
        self.steps, self.iterations = steps, iterations
        self.buffers, self.index = None, 0

    def load(self, actor, critic, norm):
        with torch.no_grad():
            for p, v in zip(self.actor_vars, actor):
                p.copy_(torch.as_tensor(v))
            for p, v in zip(self.critic_vars, critic):
                p.copy_(torch.as_tensor(v))
        self.norm_mean, self.norm_std = torch.as_tensor(norm[0]), torch.as_tensor(norm[1])

    def distribution(self, observations):
        loc = self.loc_layer(self.actor_torso(observations))
        scale = torch.nn.functional.softplus(self.log_scale) + port.FLOAT_EPSILON
        scale = torch.clamp(scale, 1e-4, 1.).repeat(observations.shape[0], 1)
        return torch.distributions.normal.Normal(loc, scale)

    def critic(self, observations):
        with torch.no_grad():
            observations = (observations - self.norm_mean) / self.norm_std
        return self.v_layer(self.critic_torso(observations)).squeeze(-1)

    def step(self, observations):                                   # a2c.py:41-52,75-85
        observations = torch.as_tensor(observations, dtype=torch.float32)
        with torch.no_grad():
            dist = self.distribution(observations)
            actions = dist.sample()
            log_probs = dist.log_prob(actions).sum(dim=-1)
        self.last = (observations.numpy().copy(), actions.numpy().copy(), log_probs.numpy().copy())
        return self.last[1]

    def store(self, next_observations, rewards, resets, terminations):   # a2c.py:58-69
        row = dict(observations=self.last[0], actions=self.last[1],
                   next_observations=next_observations, rewards=rewards, resets=resets,
                   terminations=terminations, log_probs=self.last[2])
        if self.buffers is None:
            self.buffers = {k: np.zeros((self.steps,) + np.array(v).shape, np.float32)
                            for k, v in row.items()}
        for k, v in row.items():
            self.buffers[k][self.index] = v
        self.index += 1
        self.normalizer.record(self.last[0])

    def actor_update(self, observations, actions, advantages, log_probs):   # actors.py:70-112
        self.actor_opt.zero_grad()
        dist = self.distribution(observations)
        new_log_probs = dist.log_prob(actions).sum(dim=-1)
        ratios = torch.exp(new_log_probs - log_probs)
        clipped = torch.clamp(ratios, 1 - 0.2, 1 + 0.2)
        loss = -(torch.min(advantages * ratios, advantages * clipped)).mean()
        loss.backward()
        self.actor_opt.step()
        with torch.no_grad():
            kl = (log_probs - new_log_probs).mean()
        return dict(loss=loss.detach(), kl=kl, stop=bool(kl > 0.015))

    def critic_update(self, observations, returns):                      # critics.py:18-28
        self.critic_opt.zero_grad()
        values = self.critic(observations)
        loss = torch.nn.functional.mse_loss(values, returns)
        loss.backward()
        self.critic_opt.step()
        return dict(loss=loss.detach(), v=values.detach())

    def evaluate_and_returns(self, gamma=0.99, lam=0.97):                # ppo.py:20-24
        b = self.buffers
        flat = {k: port.flatten_time_major(v) for k, v in b.items()}
        with torch.no_grad():
            values = self.critic(torch.as_tensor(flat['observations'])).numpy()
            next_values = self.critic(torch.as_tensor(flat['next_observations'])).numpy()
        shape = b['rewards'].shape
        b['values'], b['next_values'] = values.reshape(shape), next_values.reshape(shape)
        b['returns'] = port.lambda_returns(b['next_values'], b['rewards'], b['resets'],
                                           b['terminations'], gamma, lam)
        b['advantages'] = port.normalized_advantages(b['returns'], b['values'])
        self.index = 0
        return {k: torch.as_tensor(port.flatten_time_major(b[k])) for k in
                ('observations', 'actions', 'advantages', 'log_probs', 'returns')}

    def update(self, iterations=None):                                   # ppo.py:20-59
        batch = self.evaluate_and_returns()
        infos, train_actor = [], True
        for _ in range(iterations or self.iterations):
            info = {}
            if train_actor:
                info['actor'] = self.actor_update(batch['observations'], batch['actions'],
                                                  batch['advantages'], batch['log_probs'])
                train_actor = not info['actor']['stop']
            info['critic'] = self.critic_update(batch['observations'], batch['returns'])
            infos.append(info)
        mean, std = self.normalizer.update()
        self.norm_mean, self.norm_std = torch.as_tensor(mean), torch.as_tensor(std)
        return infos


class TorchTRPO(TorchPPO):
    """TRPO (``tonic/torch/agents/trpo.py:7-97``): the natural-gradient actor step of
    ``updaters/actors.py:115-156`` — surrogate ``-(ratio * advantage).mean()``, constraint
    ``KL(new || behaviour).mean()`` — solved as in ``updaters/optimizers.py:25-115``: ten conjugate
    gradient iterations on damped Fisher-vector products (double back-propagation through the KL,
    vector algebra in NumPy float32), step length ``sqrt(2 delta / x^T H x + 1e-8)``, then up to
    ten backtracking trials ``0.8 ** i`` accepted when ``KL <= delta`` and the surrogate did not
    get worse; the critic is the plain regression, ``iterations`` full-batch steps."""
    CG_STEPS, DAMPING, DELTA, BACKTRACK_STEPS, BACKTRACK = 10, 0.1, 0.01, 10, 0.8

    def _flat(self, tensors):
        return torch.cat([t.reshape(-1) for t in tensors])

    def _assign(self, flat):
        offset = 0
        with torch.no_grad():
            for p in self.actor_vars:
                p.copy_(flat[offset:offset + p.numel()].reshape(p.shape))
                offset += p.numel()

    def natural_step(self, observations, actions, log_probs, locs, scales, advantages):
        if bool((advantages == 0).all()):
            return dict(loss=0.0, kl=0.0, backtrack_steps=0)
        behaviour = torch.distributions.normal.Normal(locs, scales)

        def surrogate():
            new = self.distribution(observations).log_prob(actions).sum(dim=-1)
            return -(torch.exp(new - log_probs) * advantages).mean()

        def divergence():
            return torch.distributions.kl.kl_divergence(
                self.distribution(observations), behaviour).mean()

        def fisher_vector(v):
            first = self._flat(torch.autograd.grad(divergence(), self.actor_vars,
                                                   create_graph=True))
            second = self._flat(torch.autograd.grad((first * torch.as_tensor(v)).sum(),
                                                    self.actor_vars))
            return (second + self.DAMPING * torch.as_tensor(v)).numpy()

        start = self._flat(self.actor_vars).detach().clone()
        loss = surrogate()
        b = self._flat(torch.autograd.grad(loss, self.actor_vars)).numpy()
        start_loss = loss.detach().numpy()
        x, r, p = np.zeros_like(b), b.copy(), b.copy()
        rr = np.dot(r, r)
        if rr == 0:
            return dict(loss=0.0, kl=0.0, backtrack_steps=0)
        for _ in range(self.CG_STEPS):
            z = fisher_vector(p)
            alpha = rr / (np.dot(p, z) + port.FLOAT_EPSILON)
            x += alpha * p
            r -= alpha * z
            rr_new = np.dot(r, r)
            p = r + (rr_new / rr) * p
            rr = rr_new
        length = np.sqrt(2 * self.DELTA / np.dot(x, fisher_vector(x)) + port.FLOAT_EPSILON)
        direction = torch.as_tensor(x)

        def trial(fraction):
            self._assign(start - length * direction * fraction)
            with torch.no_grad():
                return divergence(), surrogate()

        for i in range(self.BACKTRACK_STEPS):
            kl, loss = trial(self.BACKTRACK ** i)
            if kl.numpy() <= self.DELTA and loss.numpy() <= start_loss:
                break
            if i == self.BACKTRACK_STEPS - 1:
                kl, loss = trial(0)
                i = self.BACKTRACK_STEPS
        return dict(loss=float(loss), kl=float(kl), backtrack_steps=i + 1)

    def update(self, iterations=None):                                   # trpo.py:69-97
        batch = self.evaluate_and_returns()
        if 'locs' in self.buffers:                   # as stored at acting time (trpo.py:20-33)
            locs, scales = (torch.as_tensor(port.flatten_time_major(self.buffers[k]))
                            for k in ('locs', 'scales'))
        else:                                        # the same numbers, from the unchanged policy
            with torch.no_grad():
                behaviour = self.distribution(batch['observations'])
            locs, scales = behaviour.loc, behaviour.stddev
        info = dict(actor=self.natural_step(batch['observations'], batch['actions'],
                                            batch['log_probs'], locs, scales,
                                            batch['advantages']))
        info['critic'] = [self.critic_update(batch['observations'], batch['returns'])
                          for _ in range(iterations or self.iterations)]
        mean, std = self.normalizer.update()
        self.norm_mean, self.norm_std = torch.as_tensor(mean), torch.as_tensor(std)
        return info


# ------------------------------------------------------------------- off-policy (SAC / TD3)

ACTOR_KEYS = {'sac': ('torso.model.0.weight', 'torso.model.0.bias', 'torso.model.2.weight',
                      'torso.model.2.bias', 'head.loc_layer.0.weight', 'head.loc_layer.0.bias',
                      'head.scale_layer.0.weight', 'head.scale_layer.0.bias'),
              'td3': ('torso.model.0.weight', 'torso.model.0.bias', 'torso.model.2.weight',
                      'torso.model.2.bias', 'head.action_layer.0.weight',
                      'head.action_layer.0.bias')}
ACTOR_KEYS['ddpg'] = ACTOR_KEYS['d4pg'] = ACTOR_KEYS['td3']
ACTOR_KEYS['mpo'] = ACTOR_KEYS['sac']
CRITIC_KEYS = ('torso.model.0.weight', 'torso.model.0.bias', 'torso.model.2.weight',
               'torso.model.2.bias', 'head.v_layer.weight', 'head.v_layer.bias')
DISTRIBUTIONAL_CRITIC_KEYS = CRITIC_KEYS[:4] + ('head.distributional_layer.weight',
                                                'head.distributional_layer.bias')


def project_onto_support(values, probabilities, returns):
    """``CategoricalWithSupport.project`` (tonic/torch/models/critics.py:32-46): the mass of atom j,
    moved to returns[:, j], is shared between the two support points around it (everything outside
    the support goes to the end points)."""
    vmin, vmax = values[0], values[-1]
    to_next = (torch.cat([values, vmin[None]], 0)[1:] - values)[None, :, None]
    to_previous = (values - torch.cat([vmax[None], values], 0)[:-1])[None, :, None]
    delta = torch.clamp(returns, vmin, vmax)[:, None] - values[None, :, None]
    right = (delta >= 0).float()
    distance = (right * delta / to_next) - ((1 - right) * delta / to_previous)
    return (torch.clamp(1 - distance, 0, 1) * probabilities[:, None]).sum(dim=2)


class OffPolicyPort:
    """SAC / TD3 / DDPG learner restated with torch-CPU autograd:
    ``tonic/torch/updaters/critics.py:125-134,156-182,202-235``,
    ``tonic/torch/updaters/actors.py:170-189,238-267``,
    ``tonic/torch/models/actors.py:7-34,94-98,113-115``, ``models/encoders.py:28-31``,
    ``models/actor_critics.py:126-130`` (polyak), ``tonic/replays/buffers.py:84-91`` (gather)
    and the iteration schedule of ``agents/ddpg.py:105-112`` / ``td3.py:38-47``."""

    def __init__(self, kind, state, prefix, delay_steps=2, entropy_coeff=0.2, target_coeff=0.005,
                 atoms=None, samples=20):
        """kind 'd4pg' (agents/d4pg.py, critics.py:89-122, actors.py:192-224): `atoms` =
        (vmin, vmax, count) of the DistributionalValueHead (models/critics.py:49-66).
        kind 'mpo' (agents/mpo.py, critics.py:238-282, actors.py:270-464): `samples` actions per
        state; per-dimension KL constraints and action penalisation as in the defaults."""
        self.samples = samples
        self.kind, self.delay, self.alpha, self.tau = kind, delay_steps, entropy_coeff, target_coeff
        critic_keys = DISTRIBUTIONAL_CRITIC_KEYS if kind == 'd4pg' else CRITIC_KEYS
        self.critic_keys = critic_keys
        if kind == 'd4pg':
            self.values = torch.linspace(float(atoms[0]), float(atoms[1]), int(atoms[2])).float()

        def grab(net, keys, grad):
            return [torch.tensor(state[f'{prefix}{net}.{k}'], requires_grad=grad) for k in keys]
        # DDPG (agents/ddpg.py, critics.py:56-86): ONE critic named `critic` / `target_critic`
        self.critic_names = (('critic',) if kind in ('ddpg', 'd4pg', 'mpo')
                             else ('critic_1', 'critic_2'))
        self.actor = grab('actor', ACTOR_KEYS[kind], True)
        self.critics = [grab(n, critic_keys, True) for n in self.critic_names]
        self.target_actor = grab('target_actor', ACTOR_KEYS[kind], False)
        self.target_critics = [grab('target_' + n, critic_keys, False) for n in self.critic_names]
        self.mean = torch.tensor(state[prefix + 'observation_normalizer._mean'])
        self.std = torch.tensor(state[prefix + 'observation_normalizer._std'])
        lr_actor, lr_critic = (3e-4, 3e-4) if kind in ('sac', 'mpo') else (1e-3, 1e-3)
        if kind == 'mpo':                                          # actors.py:300-316
            A = self.actor[4].shape[0]
            self.duals = [torch.nn.Parameter(torch.tensor([1.0])),
                          torch.nn.Parameter(torch.full((A,), 1.0)),
                          torch.nn.Parameter(torch.full((A,), 10.0)),
                          torch.nn.Parameter(torch.tensor([1.0]))]
            self.dual_opt = torch.optim.Adam(self.duals, lr=1e-2)
        self.actor_opt = torch.optim.Adam(self.actor, lr=lr_actor)
        self.critic_opt = torch.optim.Adam(sum(self.critics, []), lr=lr_critic)

    @staticmethod
    def torso(p, x):
        h = torch.relu(torch.nn.functional.linear(x, p[0], p[1]))
        return torch.relu(torch.nn.functional.linear(h, p[2], p[3]))

    def q(self, p, observations, actions):
        x = torch.cat([(observations - self.mean) / self.std, actions], dim=-1)
        return torch.nn.functional.linear(self.torso(p, x), p[4], p[5]).squeeze(-1)

    def policy(self, p, observations, eps):
        """Returns (actions, log_probs) — log_probs None for the deterministic head."""
        h = self.torso(p, observations)
        if self.kind != 'sac':
            return torch.tanh(torch.nn.functional.linear(h, p[4], p[5])), None
        loc = torch.nn.functional.linear(h, p[4], p[5])
        scale = torch.clamp(torch.nn.functional.softplus(
            torch.nn.functional.linear(h, p[6], p[7])), 1e-4, 1)
        raw = loc + eps * scale
        normal = torch.distributions.normal.Normal(loc, scale)
        squashed = torch.tanh(raw)
        log_probs = normal.log_prob(raw) - torch.log(1 - squashed ** 2 + 1e-6)
        return squashed, log_probs.sum(dim=-1)

    def gaussian(self, p, observations):
        """GaussianPolicyHead with its defaults (models/actors.py:69-98): tanh loc, softplus scale."""
        h = self.torso(p, observations)
        loc = torch.tanh(torch.nn.functional.linear(h, p[4], p[5]))
        scale = torch.clamp(torch.nn.functional.softplus(
            torch.nn.functional.linear(h, p[6], p[7])), 1e-4, 1)
        return loc, scale

    def sampled_values(self, observations, eps):
        """S actions per state from the target actor, valued by the target critic -> [S, B]."""
        loc, scale = self.gaussian(self.target_actor, observations)
        S = self.samples
        actions = loc + eps.view(S, -1, loc.shape[-1]) * scale
        tiled = observations[None].expand(S, *observations.shape).reshape(-1, observations.shape[-1])
        values = self.q(self.target_critics[0], tiled, actions.reshape(-1, loc.shape[-1]))
        return loc, scale, actions, values.view(S, -1)

    def mpo_actor_step(self, b, eps):
        """actors.py:318-464, restated term by term."""
        floor = torch.tensor(-18.0)
        softplus = torch.nn.functional.softplus
        with torch.no_grad():
            for d in self.duals:
                d.copy_(torch.maximum(floor, d))
            loc_t, scale_t, actions, values = self.sampled_values(b['observations'], eps)
        self.actor_opt.zero_grad()
        self.dual_opt.zero_grad()
        loc, scale = self.gaussian(self.actor, b['observations'])
        temperature = softplus(self.duals[0]) + 1e-8
        alpha_mean = softplus(self.duals[1]) + 1e-8
        alpha_std = softplus(self.duals[2]) + 1e-8
        penalty_temperature = softplus(self.duals[3]) + 1e-8
        log_samples = torch.log(torch.tensor(float(self.samples)))

        def e_step(scores, epsilon, temp):                         # actors.py:325-338
            tempered = scores / temp
            return (torch.softmax(tempered, dim=0).detach(),
                    temp * (epsilon + torch.logsumexp(tempered, dim=0).mean() - log_samples))
        weights, temperature_loss = e_step(values, 1e-1, temperature)
        outside = actions - torch.clamp(actions, -1, 1)
        penalty_weights, penalty_loss = e_step(-torch.norm(outside, dim=-1), 1e-3,
                                               penalty_temperature)
        weights = weights + penalty_weights
        temperature_loss = temperature_loss + penalty_loss
        normal = torch.distributions.normal.Normal
        fixed_std, fixed_mean, target = normal(loc, scale_t), normal(loc_t, scale), normal(loc_t, scale_t)
        policy_mean_loss = -(fixed_std.log_prob(actions).sum(-1) * weights).sum(0).mean()
        policy_std_loss = -(fixed_mean.log_prob(actions).sum(-1) * weights).sum(0).mean()
        kl_mean = torch.distributions.kl.kl_divergence(target, fixed_std).mean(0)
        kl_std = torch.distributions.kl.kl_divergence(target, fixed_mean).mean(0)
        kl_mean_loss = (alpha_mean.detach() * kl_mean).sum()
        kl_std_loss = (alpha_std.detach() * kl_std).sum()
        alpha_mean_loss = (alpha_mean * (1e-3 - kl_mean.detach())).sum()
        alpha_std_loss = (alpha_std * (1e-6 - kl_std.detach())).sum()
        loss = (policy_mean_loss + policy_std_loss + kl_mean_loss + kl_std_loss + alpha_mean_loss +
                alpha_std_loss + temperature_loss)
        loss.backward()
        self.actor_opt.step()
        self.dual_opt.step()
        scalars = dict(policy_mean_loss=policy_mean_loss, policy_std_loss=policy_std_loss,
                       kl_mean_loss=kl_mean_loss, kl_std_loss=kl_std_loss,
                       alpha_mean_loss=alpha_mean_loss, alpha_std_loss=alpha_std_loss,
                       temperature_loss=temperature_loss, temperature=temperature,
                       penalty_temperature=penalty_temperature)
        return dict({k: float(v.detach()) for k, v in scalars.items()},
                    alpha_mean=alpha_mean.detach().numpy().copy(),
                    alpha_std=alpha_std.detach().numpy().copy())

    def logits(self, p, observations, actions):
        x = torch.cat([(observations - self.mean) / self.std, actions], dim=-1)
        return torch.nn.functional.linear(self.torso(p, x), p[4], p[5])

    def critic_step(self, b, eps):
        if self.kind == 'mpo':                                     # critics.py:253-282
            with torch.no_grad():
                _, _, _, next_values = self.sampled_values(b['next_observations'], eps)
                returns = b['rewards'] + b['discounts'] * next_values.mean(dim=0)
            self.critic_opt.zero_grad()
            q = self.q(self.critics[0], b['observations'], b['actions'])
            loss = torch.nn.functional.mse_loss(returns, q)
            loss.backward()
            self.critic_opt.step()
            return dict(loss=float(loss.detach()), q1=float(q.detach().mean()), q2=0.0)
        if self.kind == 'd4pg':                                    # critics.py:100-122
            with torch.no_grad():
                next_actions, _ = self.policy(self.target_actor, b['next_observations'], None)
                next_probabilities = torch.nn.functional.softmax(self.logits(
                    self.target_critics[0], b['next_observations'], next_actions), dim=-1)
                returns = b['rewards'][:, None] + b['discounts'][:, None] * self.values
                targets = project_onto_support(self.values, next_probabilities, returns)
            self.critic_opt.zero_grad()
            log_probabilities = torch.nn.functional.log_softmax(
                self.logits(self.critics[0], b['observations'], b['actions']), dim=-1)
            loss = -(targets * log_probabilities).sum(dim=-1).mean()
            loss.backward()
            self.critic_opt.step()
            return dict(loss=float(loss.detach()), q1=0.0, q2=0.0)
        if self.kind == 'ddpg':                                    # critics.py:68-86
            with torch.no_grad():
                next_actions, _ = self.policy(self.target_actor, b['next_observations'], None)
                returns = b['rewards'] + b['discounts'] * self.q(
                    self.target_critics[0], b['next_observations'], next_actions)
            self.critic_opt.zero_grad()
            q = self.q(self.critics[0], b['observations'], b['actions'])
            loss = torch.nn.functional.mse_loss(q, returns)
            loss.backward()
            self.critic_opt.step()
            return dict(loss=float(loss.detach()), q1=float(q.detach().mean()), q2=0.0)
        with torch.no_grad():
            if self.kind == 'td3':
                next_actions, _ = self.policy(self.target_actor, b['next_observations'], None)
                noise = torch.clamp(0.2 * eps, -0.5, 0.5)
                next_actions = torch.clamp(next_actions + noise, -1, 1)
                bonus = 0
            else:
                next_actions, next_lp = self.policy(self.actor, b['next_observations'], eps)
                bonus = -self.alpha * next_lp
            next_q = torch.min(self.q(self.target_critics[0], b['next_observations'], next_actions),
                               self.q(self.target_critics[1], b['next_observations'], next_actions))
            returns = b['rewards'] + b['discounts'] * (next_q + bonus)
        self.critic_opt.zero_grad()
        q1 = self.q(self.critics[0], b['observations'], b['actions'])
        q2 = self.q(self.critics[1], b['observations'], b['actions'])
        loss = torch.nn.functional.mse_loss(q1, returns) + torch.nn.functional.mse_loss(q2, returns)
        loss.backward()
        self.critic_opt.step()
        return dict(loss=float(loss.detach()), q1=float(q1.detach().mean()),
                    q2=float(q2.detach().mean()))

    def actor_step(self, b, eps):
        if self.kind == 'mpo':
            return self.mpo_actor_step(b, eps)
        self.actor_opt.zero_grad()
        actions, lp = self.policy(self.actor, b['observations'], eps)
        if self.kind == 'd4pg':                                    # actors.py:211-214
            probabilities = torch.nn.functional.softmax(
                self.logits(self.critics[0], b['observations'], actions), dim=-1)
            loss = -(probabilities * self.values).sum(dim=-1).mean()
        elif self.kind != 'sac':
            loss = -self.q(self.critics[0], b['observations'], actions).mean()
        else:
            q = torch.min(self.q(self.critics[0], b['observations'], actions),
                          self.q(self.critics[1], b['observations'], actions))
            loss = (self.alpha * lp - q).mean()
        loss.backward()
        self.actor_opt.step()
        for p in sum(self.critics, []):
            p.grad = None
        return dict(loss=float(loss.detach()))

    def update_targets(self):
        online = self.actor + sum(self.critics, [])
        target = self.target_actor + sum(self.target_critics, [])
        with torch.no_grad():
            for o, t in zip(online, target):
                t.mul_(1 - self.tau)
                t.add_(self.tau * o)

    def update(self, buffers, workers, indices, eps):
        """indices [iterations, B]; eps [iterations, draws, B, A] (draw 0 = critic step)."""
        infos = []
        for it in range(indices.shape[0]):
            rows, cols = indices[it] // workers, indices[it] % workers
            b = {k: torch.as_tensor(buffers[k][rows, cols]) for k in (
                'observations', 'actions', 'next_observations', 'rewards', 'discounts')}
            info = dict(critic=self.critic_step(b, torch.as_tensor(eps[it, 0])))
            if self.kind != 'td3' or (it + 1) % self.delay == 0:
                actor_eps = torch.as_tensor(eps[it, 1]) if self.kind in ('sac', 'mpo') else None
                info['actor'] = self.actor_step(b, actor_eps)
                self.update_targets()
            infos.append(info)
        return infos

    def state(self):
        out = {}
        nets = [('actor', self.actor, ACTOR_KEYS[self.kind]),
                ('target_actor', self.target_actor, ACTOR_KEYS[self.kind])]
        for n, online, target in zip(self.critic_names, self.critics, self.target_critics):
            nets += [(n, online, self.critic_keys), ('target_' + n, target, self.critic_keys)]
        for net, params, keys in nets:
            for k, p in zip(keys, params):
                out[f'{net}.{k}'] = p.detach().numpy()
        return out
