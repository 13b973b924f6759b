"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the *checker*; the product path (``tonic_amd/``) never
does.  Every function cites the reference lines (relative to ``/root/reference``) whose
arithmetic it restates.  The restatement is pinned against golden vectors produced by
running the unmodified reference in the build container (``oracle/make_golden.py`` ->
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Parity status: the reference ships **no tests, fixtures or known-answer vectors**
(SURVEY.md §4/§8c), so the pin is "outputs of the reference itself run here" with
torch 2.10.0 / numpy 2.2.6, one torch thread.

All arrays are float32 unless stated; Python-float scalars multiply float32 arrays the
way NumPy-2 weak promotion does in the reference (the scalar is rounded to float32).
"""
import math

import numpy as np

F32 = np.float32
LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))   # torch/distributions/normal.py log_prob
FLOAT_EPSILON = 1e-8                              # tonic/torch/models/actors.py:4


# --------------------------------------------------------------------------- replays

def lambda_returns(next_values, rewards, resets, terminations, discount_factor,
                   trace_decay):
    """Reverse lambda-return scan — ``tonic/replays/utils.py:4-19``.

    ``values`` is only used for its shape there (``zeros_like``), so it is not an
    input here.  Inputs are ``[T, W]``; the carry starts at ``next_values[T-1]``.
    Operation order (and therefore every float32 rounding) follows lines 13-18.
    """
    steps = rewards.shape[0]
    out = np.zeros(rewards.shape, F32)
    carry = next_values[steps - 1]
    one_minus_lambda = 1 - trace_decay            # Python float, as in line 14
    for t in range(steps - 1, -1, -1):
        boot = one_minus_lambda * next_values[t] + trace_decay * carry      # :13-14
        boot = boot * (1 - resets[t])                                       # :15
        boot = boot + resets[t] * next_values[t]                            # :16
        boot = boot * (1 - terminations[t])                                 # :17
        carry = rewards[t] + discount_factor * boot                        # :18
        out[t] = carry
    return out


def lambda_returns_affine(next_values, rewards, resets, terminations, gamma, lam):
    """Affine form ``ret[t] = A[t] + B[t] * ret[t+1]`` of the same scan (SURVEY.md
    Appendix A.1) evaluated in float64 — the property the chunked HIP scan relies on.
    Used only to bound the chunk-carry reassociation error in tests."""
    nv = next_values.astype(np.float64)
    keep = (1.0 - terminations.astype(np.float64))
    cont = (1.0 - resets.astype(np.float64))
    coef_b = gamma * lam * keep * cont
    coef_a = rewards.astype(np.float64) + gamma * keep * nv * (
        cont * (1.0 - lam) + resets.astype(np.float64))
    out = np.zeros(nv.shape, np.float64)
    carry = nv[-1]
    for t in range(nv.shape[0] - 1, -1, -1):
        carry = coef_a[t] + coef_b[t] * carry
        out[t] = carry
    return out


def normalized_advantages(returns, values):
    """``Segment.get_full`` advantage block — ``tonic/replays/segments.py:41-46``:
    global (all T*W) population std, no epsilon, skipped when std == 0."""
    adv = returns - values
    std = adv.std()
    if std != 0:
        adv = (adv - adv.mean()) / std
    return adv


def flatten_time_major(x):
    """``flatten_batch`` — ``tonic/replays/utils.py:22-25`` ([T, W, ...] -> [T*W, ...])."""
    return x.reshape((x.shape[0] * x.shape[1],) + x.shape[2:])


def segment_minibatch_indices(np_random, size, batch_size, batch_iterations):
    """Index stream of ``Segment.get`` with ``batch_size`` set —
    ``tonic/replays/segments.py:58-65`` (in-place shuffle of one persistent arange)."""
    order = np.arange(size)
    for _ in range(batch_iterations):
        np_random.shuffle(order)
        for start in range(0, size, batch_size):
            yield order[start:start + batch_size].copy()


def buffer_discounts(terminations, discount_factor):
    """``Buffer.store`` discount synthesis — ``tonic/replays/buffers.py:34-36``."""
    return np.float32(1 - terminations) * discount_factor


class BufferPort:
    """``Buffer.store`` incl. ``accumulate_n_steps`` — ``tonic/replays/buffers.py:33-79``: circular
    float32 rows; with ``return_steps > 1`` every store folds the new reward / discount /
    next observation into the previous ``return_steps - 1`` rows until a reset cuts the chain."""

    KEYS = ('observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
            'discounts')

    def __init__(self, size, num_workers, return_steps=1, discount_factor=0.99):
        self.max_size = size // num_workers
        self.num_workers = num_workers
        self.return_steps = return_steps
        self.discount_factor = discount_factor
        self.buffers = None
        self.index = 0
        self.size = 0

    def store(self, **kw):
        kw = dict(kw, discounts=buffer_discounts(kw['terminations'], self.discount_factor))  # :34-36
        if self.buffers is None:                                                   # :39-45
            self.buffers = {k: np.full((self.max_size,) + np.array(kw[k]).shape, np.nan, F32)
                            for k in self.KEYS}
        for k in self.KEYS:                                                        # :48-49
            self.buffers[k][self.index] = kw[k]
        if self.return_steps > 1:                                                  # :52-53
            b = self.buffers
            rewards, next_obs, discounts = kw['rewards'], kw['next_observations'], kw['discounts']
            masks = np.ones(self.num_workers, F32)
            for i in range(min(self.size, self.return_steps - 1)):                 # :64-79
                index = (self.index - i - 1) % self.max_size
                masks = masks * (1 - b['resets'][index])
                new_rewards = b['rewards'][index] + b['discounts'][index] * rewards
                b['rewards'][index] = (1 - masks) * b['rewards'][index] + masks * new_rewards
                new_discounts = b['discounts'][index] * discounts
                b['discounts'][index] = (1 - masks) * b['discounts'][index] + masks * new_discounts
                b['next_observations'][index] = (
                    (1 - masks)[:, None] * b['next_observations'][index]
                    + masks[:, None] * next_obs)
        self.index = (self.index + 1) % self.max_size                              # :55-56
        self.size = min(self.size + 1, self.max_size)


def buffer_sample_indices(np_random, size, num_workers, batch_size):
    """``Buffer.get`` index math — ``tonic/replays/buffers.py:84-88`` (int64)."""
    flat = np_random.randint(size * num_workers, size=batch_size)
    return flat, flat // num_workers, flat % num_workers


# ------------------------------------------------------------------------ normalizer

class MeanStdPort:
    """``tonic/torch/normalizers/mean_stds.py:7-74`` restated (numpy side only)."""

    def __init__(self, shape, eps=1e-2):
        self.mean = np.zeros(shape, F32)
        self.std = np.ones(shape, F32)
        self.mean_sq = np.square(self.mean)
        self.eps = eps
        self.count = 0
        self.new_sum = 0
        self.new_sum_sq = 0
        self.new_count = 0

    def record(self, values):
        # :44-48 — one row at a time, float32 accumulators, square then add.
        for row in values:
            self.new_sum = self.new_sum + row
            self.new_sum_sq = self.new_sum_sq + np.square(row)
            self.new_count += 1

    def update(self):
        # :50-63 — Python-float weights times float32 arrays.
        total = self.count + self.new_count
        batch_mean = self.new_sum / self.new_count
        batch_mean_sq = self.new_sum_sq / self.new_count
        w_old = self.count / total
        w_new = self.new_count / total
        self.mean = w_old * self.mean + w_new * batch_mean
        self.mean_sq = w_old * self.mean_sq + w_new * batch_mean_sq
        var = np.maximum(self.mean_sq - np.square(self.mean), 0)      # :65-70
        self.std = np.maximum(np.sqrt(var), self.eps)
        self.count = total
        self.new_count = 0
        self.new_sum = 0
        self.new_sum_sq = 0
        return self.mean.astype(F32), self.std.astype(F32)

    def normalize(self, x):
        return (x - self.mean.astype(F32)) / self.std.astype(F32)     # :34-39


# ------------------------------------------------------------------------------ MLPs

def softplus(x):
    """torch.nn.functional.softplus (beta=1, threshold=20)."""
    x = np.asarray(x, F32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(F32)


def gaussian_scale(log_scale, scale_min=1e-4, scale_max=1.0):
    """``DetachedScaleGaussianPolicyHead.forward`` scale —
    ``tonic/torch/models/actors.py:63-64``.  Returns (scale, d scale / d log_scale)."""
    raw = softplus(log_scale) + F32(FLOAT_EPSILON)
    scale = np.clip(raw, F32(scale_min), F32(scale_max)).astype(F32)
    inside = (raw >= F32(scale_min)) & (raw <= F32(scale_max))
    sigmoid = (1 / (1 + np.exp(-np.asarray(log_scale, np.float64)))).astype(F32)
    return scale, np.where(inside, sigmoid, 0).astype(F32)


def torso_forward(x, w1, b1, w2, b2, activation='tanh'):
    """``MLP.forward`` — ``tonic/torch/models/utils.py:12-23`` for two hidden layers."""
    act = np.tanh if activation == 'tanh' else (lambda z: np.maximum(z, 0))
    h1 = act(x @ w1.T + b1).astype(F32)
    h2 = act(h1 @ w2.T + b2).astype(F32)
    return h1, h2


def ppo_actor_forward(params, observations):
    """``Actor.forward`` with the detached-scale Gaussian head —
    ``tonic/torch/models/actors.py:60-66,134-137``.  The actor is NOT observation-
    normalised in the torch back-end.  ``params`` is the
    reference parameter order ``[W1,b1,W2,b2,log_scale,W3,b3]`` (Appendix C)."""
    w1, b1, w2, b2, log_scale, w3, b3 = params
    h1, h2 = torso_forward(observations, w1, b1, w2, b2)
    loc = np.tanh(h2 @ w3.T + b3).astype(F32)
    scale, dscale = gaussian_scale(log_scale)
    return h1, h2, loc, scale.reshape(-1), dscale.reshape(-1)


def normal_log_prob(actions, loc, scale):
    """``torch.distributions.Normal.log_prob`` summed over the action axis
    (``a2c.py:84``, ``updaters/actors.py:82``)."""
    var = scale * scale
    per_dim = -((actions - loc) ** 2) / (2 * var) - np.log(scale) - F32(LOG_SQRT_2PI)
    return per_dim.astype(F32).sum(-1, dtype=F32)


def ppo_act(params, observations, eps):
    """``A2C._step`` — ``tonic/torch/agents/a2c.py:75-85`` with the standard-normal
    draw ``eps`` supplied (``Normal.sample()`` == ``loc + scale*eps``, Appendix A.7)."""
    _, _, loc, scale, _ = ppo_actor_forward(params, observations)
    actions = (loc + scale * eps).astype(F32)
    return actions, normal_log_prob(actions, loc, scale)


def clipped_ratio_grads(params, observations, actions, advantages, old_log_probs,
                        ratio_clip=0.2, entropy_coeff=0.0, kl_threshold=0.015, plain=False):
    """Loss, statistics and parameter gradients of ``ClippedRatio.__call__`` —
    ``tonic/torch/updaters/actors.py:70-112`` — by explicit back-propagation
    (Appendix A.3).  Returns (grads in parameter order, stats dict).  ``plain=True``:
    ``StochasticPolicyGradient.__call__`` (``actors.py:20-51``), loss = -mean(adv * logp)."""
    w1, b1, w2, b2, log_scale, w3, b3 = params
    n, a_dim = actions.shape
    h1, h2, loc, scale, dscale_dls = ppo_actor_forward(params, observations)
    var = scale * scale
    new_lp = normal_log_prob(actions, loc, scale)
    ratio = np.exp(new_lp - old_log_probs).astype(F32)
    low, high = F32(1 - ratio_clip), F32(1 + ratio_clip)
    surr1 = advantages * ratio
    surr2 = advantages * np.clip(ratio, low, high)
    loss = -np.minimum(surr1, surr2).mean(dtype=np.float64)
    if plain:
        loss = -(advantages * new_lp).mean(dtype=np.float64)
    entropy = float(np.mean(0.5 + 0.5 * math.log(2 * math.pi) + np.log(scale.astype(np.float64))))
    if entropy_coeff != 0:
        loss -= entropy_coeff * entropy
    kl = float((old_log_probs - new_lp).mean(dtype=np.float64))
    clipped = (ratio > high) | (ratio < low)
    # d loss / d new_lp: zero where the clipped surrogate is the active minimum.
    dead = ((ratio > high) & (advantages > 0)) | ((ratio < low) & (advantages < 0))
    g_lp = np.where(dead, 0, -(advantages * ratio) / n).astype(F32)
    if plain:
        g_lp = (-advantages / n).astype(F32)
    diff = actions - loc
    d_loc = g_lp[:, None] * diff / var
    d_scale = (g_lp[:, None] * (diff * diff / (var * scale) - 1 / scale)).sum(0, dtype=np.float64)
    if entropy_coeff != 0:
        d_scale = d_scale - entropy_coeff / (a_dim * scale.astype(np.float64))
    g_log_scale = (d_scale * dscale_dls).astype(F32).reshape(1, a_dim)
    d_zl = (d_loc * (1 - loc * loc)).astype(F32)
    g_w3 = d_zl.T @ h2
    g_b3 = d_zl.sum(0)
    d_z2 = ((d_zl @ w3) * (1 - h2 * h2)).astype(F32)
    g_w2 = d_z2.T @ h1
    g_b2 = d_z2.sum(0)
    d_z1 = ((d_z2 @ w2) * (1 - h1 * h1)).astype(F32)
    g_w1 = d_z1.T @ observations
    g_b1 = d_z1.sum(0)
    grads = [g_w1, g_b1, g_w2, g_b2, g_log_scale, g_w3, g_b3]
    stats = dict(loss=F32(loss), kl=F32(kl), entropy=F32(entropy),
                 clip_fraction=F32(clipped.mean()), std=F32(scale.mean()),
                 stop=bool(F32(kl) > kl_threshold))
    return [g.astype(F32) for g in grads], stats


def clip_grad_norm(grads, max_norm):
    """``torch.nn.utils.clip_grad_norm_`` (called at ``updaters/actors.py:96-98``,
    ``critics.py:24-25``): norm of the per-tensor norms, factor ``max_norm / (norm + 1e-6)``
    clamped to 1, applied to every gradient."""
    norms = np.array([np.sqrt(np.sum(np.square(g, dtype=F32), dtype=F32)) for g in grads], F32)
    total = F32(np.sqrt(np.sum(np.square(norms), dtype=F32)))
    coef = min(F32(max_norm) / (total + F32(1e-6)), F32(1.0))
    return [(g * F32(coef)).astype(F32) for g in grads], total


def critic_forward(params, mean, std, observations, clip=None):
    """``Critic.forward`` with ``ObservationEncoder`` + ``MeanStd`` + ``ValueHead`` —
    ``tonic/torch/models/critics.py:15-20,87-90``, ``encoders.py:13-16``,
    ``normalizers/mean_stds.py:34-39``.  params = ``[W1,b1,W2,b2,w3,b3]``."""
    w1, b1, w2, b2, w3, b3 = params
    x = ((observations - mean) / std).astype(F32)
    if clip is not None:                                  # mean_stds.py:37-38
        x = np.clip(x, -F32(clip), F32(clip))
    h1, h2 = torso_forward(x, w1, b1, w2, b2)
    values = (h2 @ w3.T + b3).reshape(-1).astype(F32)
    return x, h1, h2, values


def value_regression_grads(params, mean, std, observations, returns, clip=None):
    """``VRegression.__call__`` — ``tonic/torch/updaters/critics.py:18-28``:
    MSE loss and gradients; also returns the pre-step values (``v`` info)."""
    w1, b1, w2, b2, w3, b3 = params
    n = observations.shape[0]
    x, h1, h2, values = critic_forward(params, mean, std, observations, clip)
    err = values - returns
    loss = np.mean(np.square(err, dtype=np.float64))
    d_v = (2 * err / n).astype(F32)
    g_w3 = (d_v[None, :] @ h2)
    g_b3 = d_v.sum(keepdims=True)
    d_z2 = ((d_v[:, None] * w3) * (1 - h2 * h2)).astype(F32)
    g_w2 = d_z2.T @ h1
    g_b2 = d_z2.sum(0)
    d_z1 = ((d_z2 @ w2) * (1 - h1 * h1)).astype(F32)
    g_w1 = d_z1.T @ x
    g_b1 = d_z1.sum(0)
    grads = [g_w1, g_b1, g_w2, g_b2, g_w3, g_b3]
    return [g.astype(F32) for g in grads], dict(loss=F32(loss), v=values)


class AdamPort:
    """``torch.optim.Adam`` single-tensor CPU path —
    ``torch/optim/adam.py:395-547`` (betas 0.9/0.999, eps 1e-8, no weight decay),
    as constructed at ``tonic/torch/updaters/actors.py:58-59`` / ``critics.py:9-10``."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.exp_avg = [np.zeros_like(p) for p in params]
        self.exp_avg_sq = [np.zeros_like(p) for p in params]

    def step(self, params, grads):
        beta1, beta2 = self.betas
        self.step_count += 1
        bias1 = 1 - beta1 ** self.step_count
        bias2_sqrt = (1 - beta2 ** self.step_count) ** 0.5
        step_size = self.lr / bias1
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            m, v = self.exp_avg[i], self.exp_avg_sq[i]
            m = (m + F32(1 - beta1) * (g - m)).astype(F32)                    # lerp_
            v = (v * F32(beta2) + F32(1 - beta2) * g * g).astype(F32)        # mul_.addcmul_
            denom = (np.sqrt(v) / F32(bias2_sqrt) + F32(self.eps)).astype(F32)
            p = (p + F32(-step_size) * (m / denom)).astype(F32)             # addcdiv_
            self.exp_avg[i], self.exp_avg_sq[i] = m, v
            out.append(p)
        return out


def polyak(targets, onlines, coeff=0.005):
    """``update_targets`` — ``tonic/torch/models/actor_critics.py:126-130``:
    ``t.mul_(1-c); t.add_(c*o)`` (three roundings, no FMA)."""
    keep, mix = F32(1 - coeff), F32(coeff)
    return [((t * keep).astype(F32) + (mix * o).astype(F32)).astype(F32)
            for t, o in zip(targets, onlines)]


# ------------------------------------------------------------------ whole PPO update

def ppo_update(actor_params, critic_params, normalizer, segment, batch_iterations=80,
               discount_factor=0.99, trace_decay=0.97, actor_lr=3e-4, critic_lr=1e-3,
               actor_adam=None, critic_adam=None, batch_size=None, np_random=None,
               actor_clip=0, critic_clip=0, normalizer_clip=None):
    """``PPO._update`` — ``tonic/torch/agents/ppo.py:20-59`` (default full-batch path,
    ``Segment.batch_size=None``).  ``segment`` maps the seven stored keys to ``[T, W,
    ...]`` float32 arrays.  Returns new params, per-iteration infos and the returns."""
    mean, std = normalizer
    flat = {k: flatten_time_major(v) for k, v in segment.items()}
    shape = segment['rewards'].shape
    values = critic_forward(critic_params, mean, std, flat['observations'], normalizer_clip)[3]
    next_values = critic_forward(critic_params, mean, std, flat['next_observations'],
                                 normalizer_clip)[3]
    returns = lambda_returns(next_values.reshape(shape), segment['rewards'],
                             segment['resets'], segment['terminations'],
                             discount_factor, trace_decay)
    advantages = normalized_advantages(returns, values.reshape(shape)).reshape(-1)
    actor_adam = actor_adam or AdamPort(actor_params, actor_lr)
    critic_adam = critic_adam or AdamPort(critic_params, critic_lr)
    infos, train_actor = [], True
    n = advantages.shape[0]
    if batch_size is None:                      # segments.py:55-57: the same full batch each time
        schedule = [None] * batch_iterations
    else:                                       # segments.py:58-65: shuffled minibatches
        schedule = list(segment_minibatch_indices(np_random, n, batch_size, batch_iterations))
    flat_returns = returns.reshape(-1)
    for idx in schedule:
        pick = (lambda v: v) if idx is None else (lambda v: v[idx])
        obs_b, act_b, adv_b = pick(flat['observations']), pick(flat['actions']), pick(advantages)
        lp_b, ret_b = pick(flat['log_probs']), pick(flat_returns)
        info = {}
        if train_actor:
            if np.all(adv_b == 0):                             # actors.py:71-78
                _, _, _, scale, _ = ppo_actor_forward(actor_params, obs_b)
                ent = F32(np.mean(0.5 + 0.5 * math.log(2 * math.pi) + np.log(scale)))
                info['actor'] = dict(loss=F32(0), kl=F32(0), entropy=ent,
                                     clip_fraction=F32(0), std=F32(scale.mean()), stop=False)
            else:
                grads, stats = clipped_ratio_grads(actor_params, obs_b, act_b, adv_b, lp_b)
                if actor_clip > 0:
                    grads, _ = clip_grad_norm(grads, actor_clip)
                actor_params = actor_adam.step(actor_params, grads)
                info['actor'] = stats
            train_actor = not info['actor']['stop']
        grads, stats = value_regression_grads(critic_params, mean, std, obs_b, ret_b,
                                              normalizer_clip)
        if critic_clip > 0:
            grads, _ = clip_grad_norm(grads, critic_clip)
        critic_params = critic_adam.step(critic_params, grads)
        info['critic'] = dict(loss=stats['loss'], v=stats['v'])
        infos.append(info)
    return actor_params, critic_params, infos, dict(
        values=values, next_values=next_values, returns=returns, advantages=advantages)
