// Off-policy learner path (SAC / TD3) for 2-hidden-layer ReLU networks, gfx950.
//
// Restates (paths relative to the reference checkout):
//   tonic/torch/updaters/critics.py:125-134 (TargetActionNoise), :156-182
//     (TwinCriticDeterministicQLearning), :202-235 (TwinCriticSoftQLearning),
//   tonic/torch/updaters/actors.py:170-189 (DeterministicPolicyGradient), :238-267
//     (TwinCriticSoftDeterministicPolicyGradient),
//   tonic/torch/models/actors.py:7-34 (SquashedMultivariateNormalDiag), :94-98
//     (GaussianPolicyHead.forward), :113-115 (DeterministicPolicyHead), critics.py:15-20,
//     encoders.py:28-31 (ObservationActionEncoder), normalizers/mean_stds.py:34-39,
//   tonic/replays/buffers.py:33-56 (store), :81-91 (get: rows = idx // W, cols = idx % W).
//
// Every dense layer is one gemm16 launch (gemm16.h) with bias / ReLU / ReLU-mask / bias-
// gradient fused in its epilogue; the twin critics run as batch-2 launches over a flat
// [critic_1 | critic_2] parameter block.  The glue (sampling, squashing, targets, loss
// gradients) is a handful of tiny element-wise kernels.  Gradients are written as SUMS over the
// batch straight into the flat gradient buffer (same layout as the parameters), so
// tonic_adam_step / an RCCL all-reduce consume them exactly like the PPO path.
#include "gemm16.h"
#include "mlpfwd.h"
#include "collector_q.h"
#include "bufstore.h"

namespace tonic {

// Torso: MLP((H, H2), activation) of tonic/torch/models/utils.py:4-23.  The fused kernels (mlpfwd.hip) hold the
// reference's shape — two ReLU layers of one width (H2 = 0 = "as H", act = ACT_RELU); every other two-layer torso
// (unequal widths: the (400, 300) class; Tanh; ELU) runs layer by layer on gemm16 launches — `plain()` tells.
struct ActorShape {                              // heads: 1 = deterministic (TD3), 2 = loc+scale (SAC)
  int O, H, A, heads; int H2 = 0; int act = ACT_RELU;
  __host__ __device__ int h2() const { return H2 > 0 ? H2 : H; }
  __host__ __device__ int hp() const { return weight_ld(H > h2() ? H : h2()); }   // pitch of every hidden array
  bool plain() const { return h2() == H && act == ACT_RELU; }
};
struct CriticShape {
  int O, A, H; int H2 = 0; int act = ACT_RELU;
  __host__ __device__ int h2() const { return H2 > 0 ? H2 : H; }
  __host__ __device__ int hp() const { return weight_ld(H > h2() ? H : h2()); }
  bool plain() const { return h2() == H && act == ACT_RELU; }
};

// The C ABI passes ONE int32 `H`: the width of a plain torso (any width: bit 30 clear), or tonic_mlp_hidden(H1,
// H2, activation) = 1 << 30 | H1 | H2 << 12 | activation << 24 (widths below 4096; activation: GemmAct).
constexpr int32_t kHiddenPacked = 1 << 30;
struct Hidden { int H1, H2, act; };
inline Hidden unpack_hidden(int32_t code) {
  if ((code & kHiddenPacked) == 0) return Hidden{code, code, ACT_RELU};
  Hidden h{code & 4095, (code >> 12) & 4095, (code >> 24) & 7};
  if (h.H2 == 0) h.H2 = h.H1;
  if (h.act == 0) h.act = ACT_RELU;
  return h;
}
inline ActorShape actor_shape(int O, int32_t code, int A, int heads) {
  const Hidden h = unpack_hidden(code);
  return ActorShape{O, h.H1, A, heads, h.H2 == h.H1 ? 0 : h.H2, h.act};
}
inline CriticShape critic_shape(int O, int A, int32_t code) {
  const Hidden h = unpack_hidden(code);
  return CriticShape{O, A, h.H1, h.H2 == h.H1 ? 0 : h.H2, h.act};
}
inline int hidden_pitch(int32_t code) {
  const Hidden h = unpack_hidden(code);
  return weight_ld(h.H1 > h.H2 ? h.H1 : h.H2);
}
inline bool hidden_plain(int32_t code) {
  const Hidden h = unpack_hidden(code);
  return h.H1 == h.H2 && h.act == ACT_RELU;
}

// Floats of one network in the padded layout (mlpfwd.h: weight_ld / slot4):
//   actor : W1 [H, O] b1 [H] W2 [H2, H] b2 [H2] then per head Wh [A, H2] bh [A]
//   critic: W1 [H, O + A] b1 [H] W2 [H2, H] b2 [H2] w3 [1, H2] b3 [1]
__host__ __device__ inline int64_t actor_count(ActorShape s) {
  return (int64_t)s.H * weight_ld(s.O) + slot4(s.H) + (int64_t)s.h2() * weight_ld(s.H) + slot4(s.h2()) +
         (int64_t)s.heads * ((int64_t)s.A * weight_ld(s.h2()) + slot4(s.A));
}
__host__ __device__ inline int64_t critic_count(CriticShape s) {
  return (int64_t)s.H * weight_ld(s.O + s.A) + slot4(s.H) + (int64_t)s.h2() * weight_ld(s.H) +
         slot4(s.h2()) + weight_ld(s.h2()) + slot4(1);
}

// ------------------------------------------------------------------ element-wise kernels

// X[m] = [ (obs[m] - mean) / std , actions[m] ]   (encoders.py:28-31 + mean_stds.py:36)
// blockIdx.y == 1 encodes a second (obs, act) pair into X2 in the same launch.
__global__ void encode_kernel(const float* obs, const float* act, const float* mean,
                              const float* std, float clip, float* X, int B, int O, int A, int ldx,
                              const float* obs2 = nullptr, const float* act2 = nullptr,
                              float* X2 = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * (O + A)) return;
  if (blockIdx.y == 1) { obs = obs2; act = act2; X = X2; }
  const int m = idx / (O + A), c = idx - m * (O + A);
  X[(int64_t)m * ldx + c] =
      c < O ? __builtin_amdgcn_fmed3f((obs[(int64_t)m * O + c] - mean[c]) / std[c], -clip, clip)
            : act[(int64_t)m * A + (c - O)];
}

// SAC: u = loc + sigma * eps, a = tanh(u), logp = sum_a [N(u; loc, sigma) - log(1 - a^2 + 1e-6)]
// with sigma = clamp(softplus(spre), 1e-4, 1) (actors.py:11-16,94-98).  A group of G = 2^k >= A
// lanes (G <= 32; wider heads loop) owns one sample and folds the log-probability terms with a
// fixed xor tree, so the dozen transcendental calls per action run in parallel, not in a loop.
__global__ void sac_sample_kernel(const float* loc, const float* spre, const float* eps, int ld,
                                  float* act, float* logp, float* sigma_out, int B, int A,
                                  int G) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = tid / G, lane_a = tid - m * G;
  const bool sample_ok = m < B;
  float lp = 0.f;
  for (int a = lane_a; a < A; a += G) {
    if (!sample_ok) break;
    const SquashedSample sm = squashed_sample(loc[(int64_t)m * ld + a], spre[(int64_t)m * ld + a],
                                              eps ? eps[(int64_t)m * A + a] : 0.f, eps != nullptr);
    lp += sm.logp_term;
    act[(int64_t)m * A + a] = sm.action;
    if (sigma_out) sigma_out[(int64_t)m * A + a] = sm.sigma;
  }
  for (int off = G >> 1; off >= 1; off >>= 1) lp += __shfl_xor(lp, off, 64);
  if (logp && sample_ok && lane_a == 0) logp[m] = lp;
}

// TD3 target actions: clamp(a + clamp(scale * eps, -clip, clip), -1, 1)  (critics.py:130-134)
__global__ void td3_target_action_kernel(const float* loc, int ld, const float* eps, float* act,
                                         int B, int A, float scale, float clip) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int m = idx / A, a = idx - m * A;
  act[idx] = noisy_target_action(loc[(int64_t)m * ld + a], eps[idx], scale, clip);
}

// dense actions out of a padded head buffer (deterministic policy: tanh already applied)
__global__ void copy_actions_kernel(const float* loc, int ld, float* act, int B, int A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  act[idx] = loc[(int64_t)(idx / A) * ld + (idx % A)];
}

// y = r + disc * (min(q1', q2') - alpha * logp')  (critics.py:219-221; alpha = 0 and logp = null
// give TD3's critics.py:166-167), then dq_z = 2 (q_z - y) and the statistics.
// nets == 1 (DDPG, critics.py:72-79): y = r + disc * q', one critic, statistics {sq_err, q, 0}.
__global__ void critic_loss_kernel(const float* rewards, const float* discounts,
                                   const float* tq, const float* logp_next, float alpha,
                                   const float* q, float* dq, float* stats, int B, int Bp,
                                   int nets) {
  float s_loss = 0.f, s_q1 = 0.f, s_q2 = 0.f;
  for (int m = threadIdx.x; m < B; m += blockDim.x) {
    const float y = td_target(rewards, discounts, tq, logp_next, alpha, m, ValueLines{Bp, 16}, nets);
    if (nets == 1) {
      const float e1 = q[m] - y;
      dq[m] = 2.f * e1;
      s_loss += e1 * e1;
      s_q1 += q[m];
      continue;
    }
    const float e1 = q[m] - y, e2 = q[Bp + m] - y;
    dq[m] = 2.f * e1;
    dq[Bp + m] = 2.f * e2;
    s_loss += e1 * e1 + e2 * e2;
    s_q1 += q[m];
    s_q2 += q[Bp + m];
  }
  double a = s_loss, b = s_q1, c = s_q2;
  block_sum3(a, b, c);
  if (threadIdx.x == 0) {
    stats[0] = (float)a; stats[1] = (float)b; stats[2] = (float)c; stats[3] = 0.f;
    stats[4] = 0.f; stats[5] = (float)B; stats[6] = 0.f; stats[7] = 0.f;
  }
}

// Actor objective: SAC  loss = mean(alpha * logp - min(q1, q2))  (actors.py:254-257)
//                  TD3  loss = -mean(q1)                          (actors.py:177-179)
// d loss / d q_z (unscaled by 1/B): -1 on the smaller critic, -1/2 each on ties.
__global__ void actor_loss_kernel(const float* q, const float* logp, float alpha, int twin,
                                  float* dq, float* stats, int B, int Bp) {
  float s = 0.f;
  for (int m = threadIdx.x; m < B; m += blockDim.x) {
    const float q1 = q[m];
    dq[m] = actor_dq(q, m, ValueLines{Bp, 16}, twin, 0);
    if (twin) {
      dq[Bp + m] = actor_dq(q, m, ValueLines{Bp, 16}, twin, 1);
      s += alpha * logp[m] - fminf(q1, q[Bp + m]);
    } else {
      s += -q1;
    }
  }
  double a = s, unused1 = 0, unused2 = 0;
  block_sum3(a, unused1, unused2);
  if (threadIdx.x == 0) {
    stats[0] = (float)a;
    for (int i = 1; i < 8; ++i) stats[i] = i == 5 ? (float)B : 0.f;
  }
}

// ---- distributional critic (models/critics.py:23-66, D4PG): one wave per sample, lane = atom.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// softmax over the NA lanes of a wave (torch.nn.functional.softmax: exp(x - max) / sum)
__device__ __forceinline__ float wave_softmax(float logit, bool live, float* log_norm) {
  const float mx = wave_max(live ? logit : -INFINITY);
  const float e = live ? expf(logit - mx) : 0.f;
  const float sum = wave_sum(e);
  if (log_norm != nullptr) *log_norm = mx + logf(sum);
  return e / sum;
}

// DistributionalDeterministicQLearning (critics.py:100-122):
//   returns_j = r + disc * z_j ; targets = CategoricalWithSupport.project(returns) of the TARGET
//   critic's distribution (critics.py:32-46, restated with the same float32 expressions) ;
//   loss = -sum_i targets_i log_softmax(logits)_i ; d loss / d logits_i = softmax_i sum_k targets_k - targets_i
// (gradient of the SUM over the batch; the optimizer step divides by B).  loss_m: per-sample losses.
__global__ __launch_bounds__(256) void distributional_critic_loss_kernel(
    const float* target_logits, const float* logits, int ld, const float* rewards,
    const float* discounts, const float* values, int NA, float* dlogits, float* loss_m, int B) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= B) return;
  const bool live = lane < NA;
  const int i = live ? lane : NA - 1;
  const float z = values[i], vmin = values[0], vmax = values[NA - 1];
  const float p_next = wave_softmax(target_logits[(int64_t)m * ld + i], live, nullptr);
  const float ret = rewards[m] + discounts[m] * z;
  const float clipped = fminf(fmaxf(ret, vmin), vmax);
  const float d_pos = (i + 1 < NA ? values[i + 1] : vmin) - z;      // critics.py:34-35
  const float d_neg = z - (i > 0 ? values[i - 1] : vmax);           // critics.py:36-37
  float target = 0.f;
  for (int j = 0; j < NA; ++j) {
    const float delta = __shfl(clipped, j, 64) - z;                 // critics.py:40
    const float sign = delta >= 0.f ? 1.f : 0.f;
    const float hat = (sign * delta / d_pos) - ((1.f - sign) * delta / d_neg);
    target += fminf(fmaxf(1.f - hat, 0.f), 1.f) * __shfl(p_next, j, 64);
  }
  if (!live) target = 0.f;
  float log_norm;
  const float logit = logits[(int64_t)m * ld + i];
  const float p = wave_softmax(logit, live, &log_norm);
  const float total = wave_sum(target);
  const float loss = -wave_sum(live ? target * (logit - log_norm) : 0.f);
  if (lane < ld) dlogits[(int64_t)m * ld + lane] = live ? p * total - target : 0.f;
  if (lane == 0) loss_m[m] = loss;
}

// DistributionalDeterministicPolicyGradient (actors.py:203-224): value = sum_i softmax(logits)_i z_i,
// loss = -mean(value): d(-value) / d logits_i = -p_i (z_i - value).
__global__ __launch_bounds__(256) void distributional_actor_loss_kernel(
    const float* logits, int ld, const float* values, int NA, float* dlogits, float* loss_m,
    int B) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= B) return;
  const bool live = lane < NA;
  const int i = live ? lane : NA - 1;
  const float z = values[i];
  const float p = wave_softmax(logits[(int64_t)m * ld + i], live, nullptr);
  const float value = wave_sum(live ? p * z : 0.f);
  if (lane < ld) dlogits[(int64_t)m * ld + lane] = live ? -(p * (z - value)) : 0.f;
  if (lane == 0) loss_m[m] = -value;
}

// the 8 statistics {loss_sum, 0, 0, 0, 0, B, 0, 0} from per-sample losses (fixed order, float64)
__global__ void loss_stats_kernel(const float* loss_m, float* stats, int B) {
  double a = 0, unused1 = 0, unused2 = 0;
  for (int m = threadIdx.x; m < B; m += blockDim.x) a += loss_m[m];
  block_sum3(a, unused1, unused2);
  if (threadIdx.x == 0) {
    stats[0] = (float)a;
    for (int i = 1; i < 8; ++i) stats[i] = i == 5 ? (float)B : 0.f;
  }
}

// ---- MPO (agents/mpo.py; updaters/critics.py:238-282, updaters/actors.py:270-464)
// GaussianPolicyHead (models/actors.py:69-98): loc = tanh(.) (applied by the forward), sigma =
// clamp(softplus(spre), 1e-4, 1).
__device__ __forceinline__ float gaussian_sigma(float spre) {
  return fminf(fmaxf(softplus_f(spre), 1e-4f), 1.0f);
}

// Normal.sample / rsample: a = loc + sigma * eps (eps == null: the greedy loc, mpo.py:82-85)
__global__ void gaussian_sample_kernel(const float* loc, const float* spre, const float* eps,
                                       int ld, float* act, int B, int A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int m = idx / A, a = idx - m * A;
  const float l = loc[(int64_t)m * ld + a];
  act[idx] = eps ? l + gaussian_sigma(spre[(int64_t)m * ld + a]) * eps[idx] : l;
}

// S samples per state, tiled like updaters.tile + merge_first_two_dims (row s * B + m):
//   act[s, m] = loc[m] + sigma[m] * eps[s, m] ;  X[s * B + m] = [ norm(obs[m]) | act[s, m] ]
__global__ void gaussian_tile_kernel(const float* obs, const float* loc, const float* spre,
                                     const float* eps, int ldh, const float* mean,
                                     const float* std, float clip, float* act, float* X, int B,
                                     int O, int A, int S, int ldx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)S * B * (O + A)) return;
  const int64_t r = idx / (O + A);
  const int c = (int)(idx - r * (O + A));
  const int m = (int)(r % B);
  if (c < O) {
    X[r * ldx + c] = __builtin_amdgcn_fmed3f((obs[(int64_t)m * O + c] - mean[c]) / std[c], -clip, clip);
  } else {
    const int a = c - O;
    const float v = loc[(int64_t)m * ldh + a] + gaussian_sigma(spre[(int64_t)m * ldh + a]) * eps[r * A + a];
    act[r * A + a] = v;
    X[r * ldx + c] = v;
  }
}

// next_values.view(S, -1).mean(dim=0) (critics.py:269-270)
__global__ void sample_mean_kernel(const float* q, float* out, int B, int S) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= B) return;
  float sum = 0.f;
  for (int s = 0; s < S; ++s) sum += q[(int64_t)s * B + m];
  out[m] = sum / (float)S;
}

constexpr int kMpoMaxSamples = 64;
constexpr int kMpoStats = 16;    // per-state partials: see mpo_state_kernel

// duals [2 A + 2] = {log_temperature, log_alpha_mean[A], log_alpha_std[A], log_penalty_temperature}
// (actors.py:300-316, per_dim_constraining); value = softplus(log) + 1e-8 (actors.py:378-383)
__device__ __forceinline__ float dual_value(float log_dual) { return softplus_f(log_dual) + 1e-8f; }

// E-step weights and the M-step gradients at the head outputs, one WAVE per state: lane = sample for
// the weights, lane = action dimension for the gradients (actors.py:384-432).  SUMS over the batch
// (the optimizer step divides by B):
//   d/d loc   = -sum_s W_s (a_s - loc) / sigma_t^2 + alpha_mean (loc - loc_t) / sigma_t^2
//   d/d sigma = -sum_s W_s ((a_s - loc_t)^2 / sigma^3 - 1 / sigma) + alpha_std (1 / sigma - sigma_t^2 / sigma^3)
// with W = softmax_s(q / T) + softmax_s(bound cost / T_penalty).  part[m][.] = {policy_mean, policy_std,
// LSE, sum_s w q / T, LSE_penalty, sum_s w_p cost / T_p}; klm / kls [m][a] = the per-dimension KLs.
__global__ __launch_bounds__(256) void mpo_state_kernel(
    const float* q, const float* act, const float* loc_t, const float* spre_t, const float* loc,
    const float* spre, int ldh, const float* duals, float floor, int penalize, float* dloc, float* dspre,
    float* part, float* klm, float* kls, int B, int A, int S) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= B) return;
  // (every log-dual is read through the floor the reference clamps it to, in place, at the head of its call —
  //  actors.py:347-356; mpo_dual_kernel, the last reader ahead of the duals' optimizer step, writes the
  //  clamped values back)
  const float T = dual_value(fmaxf(duals[0], floor)), Tp = dual_value(fmaxf(duals[2 * A + 1], floor));
  // weights_and_temperature_loss (actors.py:325-338): softmax over the samples of q / T
  const bool sample = lane < S;
  const float tempered = sample ? q[(int64_t)lane * B + m] / T : -INFINITY;
  const float mx = wave_max(tempered);
  const float e = sample ? expf(tempered - mx) : 0.f;
  const float sum = wave_sum(e);
  const float lse = mx + logf(sum);
  float w = e / sum;
  const float wq = wave_sum(sample ? w * tempered : 0.f);
  float lse_p = 0.f, wc = 0.f;
  if (penalize) {                                              // actors.py:388-398
    float n2 = 0.f;
    if (sample) {
      for (int a = 0; a < A; ++a) {
        const float v = act[((int64_t)lane * B + m) * A + a];
        const float d = v - fminf(fmaxf(v, -1.f), 1.f);
        n2 += d * d;
      }
    }
    const float cost = sample ? -sqrtf(n2) / Tp : -INFINITY;
    const float mp = wave_max(cost);
    const float ep = sample ? expf(cost - mp) : 0.f;
    const float sp = wave_sum(ep);
    lse_p = mp + logf(sp);
    const float wp = ep / sp;
    wc = wave_sum(sample ? wp * cost : 0.f);
    w += wp;
  }
  // lane = action dimension: the sums over the samples in sample order
  const bool live = lane < A;
  const int a = live ? lane : A - 1;
  const float lt = loc_t[(int64_t)m * ldh + a], st = gaussian_sigma(spre_t[(int64_t)m * ldh + a]);
  const float lo = loc[(int64_t)m * ldh + a], pre = spre[(int64_t)m * ldh + a];
  const float sg = gaussian_sigma(pre);
  float pm = 0.f, ps = 0.f, g_loc = 0.f, g_sigma = 0.f;
  for (int s = 0; s < S; ++s) {
    const float ws = __shfl(w, s, 64);
    const float v = act[((int64_t)s * B + m) * A + a];
    const float d_mean = v - lo, d_std = v - lt;
    // Normal.log_prob: -(x - mu)^2 / (2 var) - log(sigma) - log(sqrt(2 pi))
    pm += ws * (-(d_mean * d_mean) / (2.f * (st * st)) - logf(st) - kHalfLog2Pi);
    ps += ws * (-(d_std * d_std) / (2.f * (sg * sg)) - logf(sg) - kHalfLog2Pi);
    g_loc -= ws * d_mean / (st * st);
    g_sigma -= ws * (d_std * d_std / (sg * sg * sg) - 1.f / sg);
  }
  pm = wave_sum(live ? pm : 0.f);
  ps = wave_sum(live ? ps : 0.f);
  if (live) {
    // kl_divergence(Normal(lt, st), Normal(lo, st)) and (.., Normal(lt, sg)) (actors.py:416-421)
    const float ratio = st / sg;
    klm[(int64_t)m * A + a] = 0.5f * ((lt - lo) / st) * ((lt - lo) / st);
    kls[(int64_t)m * A + a] = 0.5f * (ratio * ratio - 1.f - logf(ratio * ratio));
    const float alpha_mean = dual_value(fmaxf(duals[1 + a], floor)),
                alpha_std = dual_value(fmaxf(duals[1 + A + a], floor));
    g_loc += alpha_mean * (lo - lt) / (st * st);
    g_sigma += alpha_std * (1.f / sg - st * st / (sg * sg * sg));
    const float raw = softplus_f(pre);
    const bool inside = raw >= 1e-4f && raw <= 1.0f;
    dloc[(int64_t)m * ldh + a] = g_loc * (1.f - lo * lo);                  // tanh loc head
    dspre[(int64_t)m * ldh + a] = inside ? g_sigma / (1.f + expf(-pre)) : 0.f;
  }
  if (lane == 0) {
    float* out = part + (int64_t)m * kMpoStats;
    out[0] = pm; out[1] = ps; out[2] = lse; out[3] = wq; out[4] = lse_p; out[5] = wc;
  }
}

// Batch means -> the logged losses, the dual variables' values and their gradients (one workgroup).
// stats [7 + 2 A + 2] = {policy_mean_loss, policy_std_loss, kl_mean_loss, kl_std_loss, alpha_mean_loss,
// alpha_std_loss, temperature_loss, temperature, alpha_mean[A], alpha_std[A], penalty_temperature};
// dual_grads [2 A + 2 + 8]: d loss / d log-duals + the statistics slot of the optimizer step.
//
// Several ranks (each holds B of the B_norm states of the global batch): everything below is a
// function of the column MEANS, so the kernel runs in two halves around one all-reduce —
// out_sums != null: only the local column sums [6 + 2 A] (float64) are written; in_sums != null: the
// columns are taken from there (the all-reduced sums) instead of the per-state arrays.
__global__ void mpo_dual_kernel(const float* part, const float* klm, const float* kls,
                                float* duals, float floor, int penalize, float epsilon,
                                float epsilon_penalty, float epsilon_mean, float epsilon_std,
                                float* dual_grads, float* stats, float* actor_stats, int B, int A,
                                int S, const double* in_sums, double* out_sums, int B_norm) {
  __shared__ double col[2 * 64 + 6];
  const int tid = threadIdx.x;
  // column means in fixed order: columns 0..5 the six partials, 6..6+A-1 kl_mean, then kl_std; one
  // wave per column, the lanes stride over the batch
  const int columns = 6 + 2 * A, lane = tid & 63, waves = blockDim.x >> 6;
  for (int c = tid >> 6; c < columns; c += waves) {
    double sum = 0;
    if (in_sums != nullptr) {
      sum = in_sums[c];
    } else {
      for (int m = lane; m < B; m += 64)
        sum += c < 6 ? part[(int64_t)m * kMpoStats + c]
                     : c < 6 + A ? klm[(int64_t)m * A + (c - 6)] : kls[(int64_t)m * A + (c - 6 - A)];
      sum = wave_sum(sum);
    }
    if (lane == 0) {
      if (out_sums != nullptr) out_sums[c] = sum;
      col[c] = sum / B_norm;
    }
  }
  if (out_sums != nullptr) return;                    // (uniform)
  __syncthreads();
  if (tid >= 64) return;                              // wave 0: lane = action dimension
  const float log_T = fmaxf(duals[0], floor), log_Tp = fmaxf(duals[2 * A + 1], floor);
  const float T = dual_value(log_T), Tp = dual_value(log_Tp);
  const float log_S = logf((float)S);
  auto sigmoid = [](float x) { return 1.f / (1.f + expf(-x)); };
  float kl_mean_loss = 0.f, kl_std_loss = 0.f, alpha_mean_loss = 0.f, alpha_std_loss = 0.f;
  if (tid < A) {
    const int a = tid;
    const float log_am = fmaxf(duals[1 + a], floor), log_as = fmaxf(duals[1 + A + a], floor);
    const float am = dual_value(log_am), as = dual_value(log_as);
    const float km = (float)col[6 + a], ks = (float)col[6 + A + a];
    kl_mean_loss = am * km; kl_std_loss = as * ks;                        // actors.py:319-323
    alpha_mean_loss = am * (epsilon_mean - km);
    alpha_std_loss = as * (epsilon_std - ks);
    dual_grads[1 + a] = (epsilon_mean - km) * sigmoid(log_am);
    dual_grads[1 + A + a] = (epsilon_std - ks) * sigmoid(log_as);
    duals[1 + a] = log_am; duals[1 + A + a] = log_as;      // the reference's in-place clamp (actors.py:347-356)
    stats[8 + a] = am; stats[8 + A + a] = as;
  }
  kl_mean_loss = wave_sum(kl_mean_loss); kl_std_loss = wave_sum(kl_std_loss);
  alpha_mean_loss = wave_sum(alpha_mean_loss); alpha_std_loss = wave_sum(alpha_std_loss);
  if (tid != 0) return;
  // temperature * (epsilon + mean(logsumexp) - log S): d / dT = epsilon + mean(LSE) - log S - mean(sum_s w q / T)
  float temperature_loss = T * (epsilon + (float)col[2] - log_S);
  dual_grads[0] = (epsilon + (float)col[2] - log_S - (float)col[3]) * sigmoid(log_T);
  duals[0] = log_T;
  if (penalize) duals[2 * A + 1] = log_Tp;
  dual_grads[2 * A + 1] = 0.f;
  if (penalize) {
    temperature_loss += Tp * (epsilon_penalty + (float)col[4] - log_S);
    dual_grads[2 * A + 1] =
        (epsilon_penalty + (float)col[4] - log_S - (float)col[5]) * sigmoid(log_Tp);
  }
  for (int i = 0; i < 8; ++i) dual_grads[2 * A + 2 + i] = i == 5 ? 1.f : 0.f;
  stats[0] = -(float)col[0]; stats[1] = -(float)col[1]; stats[2] = kl_mean_loss; stats[3] = kl_std_loss;
  stats[4] = alpha_mean_loss; stats[5] = alpha_std_loss; stats[6] = temperature_loss; stats[7] = T;
  stats[8 + 2 * A] = Tp;
  // the actor's statistics slot {loss_sum = B * (policy losses + KL losses), 0, 0, 0, 0, B, 0, 0}
  actor_stats[0] = (float)B * (stats[0] + stats[1] + kl_mean_loss + kl_std_loss);
  for (int i = 1; i < 8; ++i) actor_stats[i] = i == 5 ? (float)B : 0.f;
}

// Back through the squashed Gaussian head (SAC) or the tanh head (TD3).
//   da[m][a] = d loss / d action (the action columns of the critics' input gradients, [B, pad16(A)])
// SAC: u = loc + sigma eps, a = tanh(u):
//   d loss/d u   = da (1 - a^2) + alpha * 2 a (1 - a^2) / (1 - a^2 + 1e-6)
//   d loss/d loc = d loss/d u ;  d loss/d sigma = d loss/d u * eps - alpha / sigma
//   d sigma/d spre = sigmoid(spre) inside the clamp, else 0.
// TD3: a = tanh(z):  d loss/d z = da (1 - a^2).
__global__ void actor_head_backward_kernel(const float* dxa0, const float* dxa1, int ldxa,
                                           const float* act,
                                           const float* eps, const float* sigma,
                                           const float* spre, int ld, float alpha, int sac,
                                           float* dloc, float* dspre, int B, int A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int m = idx / A, a = idx - m * A;
  // d loss / d action = sum over the critics of the action columns of their input gradient
  float da = dxa0[(int64_t)m * ldxa + a];
  if (dxa1 != nullptr) da = da + dxa1[(int64_t)m * ldxa + a];
  const float t = act[idx];
  const float one_m = 1.f - t * t;
  if (!sac) {
    dloc[(int64_t)m * ld + a] = da * one_m;
    return;
  }
  const float du = da * one_m + alpha * (2.f * t * one_m / (one_m + kSacLogEps));
  const float sg = sigma[idx];
  const float dsigma = du * eps[idx] - alpha / sg;
  const float pre = spre[(int64_t)m * ld + a];
  const float raw = softplus_f(pre);
  const bool inside = raw >= 1e-4f && raw <= 1.0f;
  dloc[(int64_t)m * ld + a] = du;
  dspre[(int64_t)m * ld + a] = inside ? dsigma / (1.f + expf(-pre)) : 0.f;
}

// Buffer.get gather (buffers.py:84-91): one wave per sampled transition.
struct GatherArgs {
  const int64_t* indices;
  const float* obs; const float* act; const float* next; const float* rew; const float* disc;
  float* o_obs; float* o_act; float* o_next; float* o_rew; float* o_disc;
  int64_t W;
  int B, O, A;
};

__global__ __launch_bounds__(256) void buffer_gather_kernel(GatherArgs g) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= g.B) return;
  const int64_t idx = g.indices[b];
  const int64_t row = idx / g.W, col = idx - row * g.W;       // rows = idx // W, cols = idx % W
  const int64_t t = row * g.W + col;
  for (int k = lane; k < g.O; k += 64) {
    g.o_obs[(int64_t)b * g.O + k] = g.obs[t * g.O + k];
    g.o_next[(int64_t)b * g.O + k] = g.next[t * g.O + k];
  }
  for (int k = lane; k < g.A; k += 64) g.o_act[(int64_t)b * g.A + k] = g.act[t * g.A + k];
  if (lane == 0) { g.o_rew[b] = g.rew[t]; g.o_disc[b] = g.disc[t]; }
}

// Buffer.accumulate_n_steps (buffers.py:58-79) after the row write of one store: thread = one
// (worker, next-observation feature) element; the reset mask chain of a worker is recomputed per
// thread (at most return_steps - 1 reads).  Every expression keeps the reference's separate
// float32 roundings — `(1 - m) * old + m * new` is NOT replaced by a select, so that even the
// sign of a zero comes out as in NumPy.
struct NStepArgs {
  float* b_next; float* b_rew; float* b_disc; const float* b_rst;
  const float* next; const float* rew; const float* term;
  int64_t row, size, max_size, W;
  int O, back;
  float discount;
};

__global__ __launch_bounds__(256) void buffer_nstep_kernel(NStepArgs a) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.W * a.O) return;
  const int64_t w = e / a.O;
  const int k = (int)(e - w * a.O);
  const float reward = a.rew[w];
  const float discount = (1.f - a.term[w]) * a.discount;             // buffers.py:34-36
  const float next = a.next[e];
  float mask = 1.f;
  for (int i = 0; i < a.back; ++i) {
    int64_t index = (a.row - i - 1) % a.max_size;
    if (index < 0) index += a.max_size;                              // Python modulo
    const int64_t at = index * a.W + w;
    mask = mask * (1.f - a.b_rst[at]);
    const float keep = 1.f - mask;
    if (k == 0) {
      const float r_old = a.b_rew[at], d_old = a.b_disc[at];
      const float scaled = d_old * reward;
      const float new_reward = r_old + scaled;
      const float r_keep = keep * r_old, r_take = mask * new_reward;
      a.b_rew[at] = r_keep + r_take;
      const float new_discount = d_old * discount;
      const float d_keep = keep * d_old, d_take = mask * new_discount;
      a.b_disc[at] = d_keep + d_take;
    }
    const float o_keep = keep * a.b_next[at * a.O + k], o_take = mask * next;
    a.b_next[at * a.O + k] = o_keep + o_take;
  }
}

// Buffer.store row write (buffers.py:33-52) + MeanStd.record (mean_stds.py:44-48): bufstore.h
__global__ __launch_bounds__(1024) void buffer_store_kernel(BufferStoreArgs a) {
  __shared__ float tile[16384];
  buffer_store_body(a, tile, 16384, (int)blockIdx.x, (int)gridDim.x);
}

// --------------------------------------------------------------------------- host helpers

namespace {

inline int pad16(int x) { return (x + 15) / 16 * 16; }
// Row pitch of the batch-major scratch matrices the weight-gradient GEMMs walk along the batch
// (hidden activations, their gradients, the critics' input): never a multiple of 128 bytes
// (profiles/r01_ubench_row_stride.md).
inline int pitch16(int x) { return weight_ld(pad16(x)); }

// Pointers into one actor's block of a flat buffer (parameters, or the gradient sums of the same
// layout); ld1 / ldH: row strides of W1 and of the H-column weights.
template <typename T>
struct ActorBlock {
  T *W1, *b1, *W2, *b2, *Wh;                    // Wh: first head; the second follows head_stride on
  ActorShape s;
  int ld1, ldH;
  int64_t head_stride;
  int ldO;                                       // row pitch of the heads (their inputs: the second layer)
  ActorBlock(T* p, ActorShape sh) : s(sh), ld1(weight_ld(sh.O)), ldH(weight_ld(sh.H)), ldO(weight_ld(sh.h2())) {
    W1 = p; b1 = W1 + (int64_t)s.H * ld1; W2 = b1 + slot4(s.H); b2 = W2 + (int64_t)s.h2() * ldH;
    Wh = b2 + slot4(s.h2());
    head_stride = (int64_t)s.A * ldO + slot4(s.A);
  }
  T* head_w(int h) const { return Wh + h * head_stride; }
  T* head_b(int h) const { return head_w(h) + (int64_t)s.A * ldO; }
};
using ActorParams = ActorBlock<const float>;

struct CriticOffsets {
  int64_t W1, b1, W2, b2, w3, b3, count;
  int ld1, ldH, ldO;                             // ldO: pitch of the value head's row (second layer wide)
  explicit CriticOffsets(CriticShape s) : ld1(weight_ld(s.O + s.A)), ldH(weight_ld(s.H)), ldO(weight_ld(s.h2())) {
    W1 = 0; b1 = W1 + (int64_t)s.H * ld1; W2 = b1 + slot4(s.H); b2 = W2 + (int64_t)s.h2() * ldH;
    w3 = b2 + slot4(s.h2()); b3 = w3 + ldO; count = b3 + slot4(1);
  }
};

GemmArgs gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
              int K) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.alpha = 1.f;
  return g;
}

#define TRY(expr)                    \
  do {                               \
    const int rc__ = (expr);         \
    if (rc__ != TONIC_OK) return rc__; \
  } while (0)

// The fp16x2 weight images of a network (mlpimg.h): `block` = this network's (the first of `nets` critics,
// the others `v.bytes` on), `target` = its polyak target's (where the optimizer epilogue keeps those up to date),
// `second` = the images of a launch's second parameter set (critics_forward's params2).  Null block = float32 passes.
struct ActorImg { char* block; char* target; ActorImages v; };
struct CriticImg { char* block; char* target; char* second; CriticImages v; };

// build_weight_images: every image of one actor / of `nets` critics from the float32 parameters
void add_actor_images(ImgBuild& b, const float* params, ActorShape s, const ActorImg& img) {
  const ActorBlock<const float> p(params, s);
  b.add(p.W1, p.ld1, s.H, s.O, img.block, img.v.f1);
  b.add(p.W2, p.ldH, s.H, s.H, img.block, img.v.f2);
  b.add(p.W2, p.ldH, s.H, s.H, img.block, img.v.t2, true, 0, s.H);
  for (int h = 0; h < s.heads; ++h) {
    b.add(p.head_w(h), p.ldO, s.A, s.H, img.block, img.v.fh[h]);
    b.add(p.head_w(h), p.ldO, s.A, s.H, img.block, img.v.th[h], true, 0, s.H);
  }
}
void add_critic_images(ImgBuild& b, const float* params, CriticShape s, int nets, char* block,
                       const CriticImages& v) {
  const CriticOffsets o(s);
  for (int z = 0; z < nets; ++z) {
    const float* p = params + z * o.count;
    char* at = block + z * v.bytes;
    b.add(p + o.W1, o.ld1, s.H, s.O + s.A, at, v.f1);
    b.add(p + o.W2, o.ldH, s.H, s.H, at, v.f2);
    b.add(p + o.W2, o.ldH, s.H, s.H, at, v.t2, true, 0, s.H);
    b.add(p + o.W1, o.ld1, s.H, s.O + s.A, at, v.t1a, true, s.O, s.A);
  }
}
int64_t images_floats(int O, int A, int H, int heads) {
  // [actor | target actor | two critics | two target critics], 256-byte slots
  return 2 * round_up(actor_images(O, H, A, heads).bytes, 256) / 4 +
         2 * round_up(2 * critic_images(O, A, H).bytes, 256) / 4;
}
bool images_serve(int O, int A, int H) {
  return g_q_images.load() != 0 && H >= 16 && H <= 256 && H % 16 == 0 && A <= 64 &&
         mlp_image_pass_supported(O + A, H);
}

// actor torso + heads: h1, h2 [Bp, H]; head h -> out_h [Bp, ldh] (pre-activation unless `tanh_head`)
// What follows the heads (sampling / target noise / dense copy); folded into the forward launch
// when the fused kernel can (then *tail_done = true), else the caller launches its own kernel.
struct PolicyTail {
  int post;                    // PolicyPost
  const float* eps;
  float* actions; float* sigma; float* logp;
  float noise_scale, noise_clip;
  // optional: the critics' input rows [normalised observations | these actions] -> enc_out, and a
  // second pair from stored actions -> enc_out2 (see MlpFwdArgs)
  const float* enc_obs; const float* enc_obs2; const float* enc_act2;
  const float* enc_mean; const float* enc_std;
  float enc_clip;
  float* enc_out; float* enc_out2;
  int enc_ld;
};

int actor_forward(const float* params, ActorShape s, const float* obs, int B, float* h1,
                  float* h2, float* head0, float* head1, int ldh, bool tanh_head,
                  hipStream_t st, const PolicyTail* tail = nullptr, bool* tail_done = nullptr,
                  int ldx = 0, const ActorImg* img = nullptr, const CollectorStep* step = nullptr,
                  float* rows_out = nullptr, const BufferStoreArgs* store = nullptr) {
  ActorParams p(params, s);
  if (ldx <= 0) ldx = s.O;                      // dense observation rows unless told otherwise
  if (tail_done != nullptr) *tail_done = false;
  const int HP = s.hp(), H2 = s.h2();
  if (s.plain() && mlp_forward_supported(s.H, s.A, s.heads)) {      // one launch for torso + heads
    MlpFwdArgs f{};
    f.X = obs; f.ldx = ldx; f.K1 = s.O;
    f.W1 = p.W1; f.b1 = p.b1; f.W2 = p.W2; f.b2 = p.b2; f.ldw1 = p.ld1; f.ldw2 = p.ldH;
    f.Wh[0] = p.head_w(0); f.bh[0] = p.head_b(0);
    f.Wh[1] = p.head_w(s.heads - 1); f.bh[1] = p.head_b(s.heads - 1);
    f.heads = s.heads; f.NH = s.A;
    f.h1 = h1; f.h2 = h2; f.ldh = HP;
    f.out[0] = head0; f.out[1] = s.heads == 2 ? head1 : head0; f.ldo = ldh;
    f.act[0] = tanh_head ? ACT_TANH : ACT_NONE; f.act[1] = ACT_NONE;
    f.B = B; f.H = s.H; f.split = 1 << 30;
    if (img != nullptr && img->block != nullptr)
      f.img = FwdImages{img->block, img->v.f1, img->v.f2, {img->v.fh[0], img->v.fh[s.heads - 1]}, 0, 0};
    if (step != nullptr) {       // a step of a collector's block: completion words, the rows' device copy
      f.done_flags = step->done_flags; f.done_seq = step->seq; f.rows_out = rows_out; f.rows_ld = s.O;
      if (store != nullptr) { f.store = *store; f.store_on = 1; }
    }
    if (tail != nullptr && g_policy_tail != 0 && mlp_policy_tail_supported(s.H, s.A)) {
      f.post = tail->post; f.post_eps = tail->eps; f.post_actions = tail->actions;
      f.post_sigma = tail->sigma; f.post_logp = tail->logp;
      f.noise_scale = tail->noise_scale; f.noise_clip = tail->noise_clip;
      f.enc_obs = tail->enc_obs; f.enc_obs2 = tail->enc_obs2; f.enc_act2 = tail->enc_act2;
      f.enc_mean = tail->enc_mean; f.enc_std = tail->enc_std; f.enc_clip = tail->enc_clip;
      f.enc_out = tail->enc_out; f.enc_out2 = tail->enc_out2; f.enc_O = s.O; f.enc_ld = tail->enc_ld;
      if (tail_done != nullptr) *tail_done = true;
    }
    return launch_mlp_forward(f, 1, st);
  }
  GemmArgs g = gemm(obs, ldx, p.W1, p.ld1, h1, HP, B, s.H, s.O);
  g.bias = p.b1; g.act = s.act;
  TRY(launch_gemm('c', 'c', g, 1, st));
  g = gemm(h1, HP, p.W2, p.ldH, h2, HP, B, H2, s.H);
  g.bias = p.b2; g.act = s.act;
  TRY(launch_gemm('c', 'c', g, 1, st));
  g = gemm(h2, HP, p.head_w(0), p.ldO, head0, ldh, B, s.A, H2);
  g.bias = p.head_b(0); g.act = tanh_head ? ACT_TANH : ACT_NONE;
  TRY(launch_gemm('c', 'c', g, 1, st));
  if (s.heads == 2) {
    g = gemm(h2, HP, p.head_w(1), p.ldO, head1, ldh, B, s.A, H2);
    g.bias = p.head_b(1);
    TRY(launch_gemm('c', 'c', g, 1, st));
  }
  return TONIC_OK;
}

// `nets` critics batched over blockIdx.z; X shared ([Bp, ldx]); h1/h2 [nets][Bp][H]; q [nets][Bp]
// params2 / X2 != null: `nets` more critics (e.g. the online ones next to the targets) with their
// own input, appended to the same launch: h1/h2 [2 nets][Bp][H], q [2 nets][Bp].
// The fused form's arguments (mlp_forward_supported(H, 1, 1)); *launch_nets = networks in the launch.
MlpFwdArgs critics_forward_args(const float* params, CriticShape s, int nets, const float* X,
                                int ldx, int B, int Bp, float* h1, float* h2, float* q,
                                const float* params2, const float* X2, int* launch_nets,
                                const CriticImg* img = nullptr) {
  const CriticOffsets o(s);
  const int HP = weight_ld(s.H);
  MlpFwdArgs f{};
  f.X = X; f.ldx = ldx; f.K1 = s.O + s.A;
  f.W1 = params + o.W1; f.b1 = params + o.b1; f.W2 = params + o.W2; f.b2 = params + o.b2;
  f.ldw1 = o.ld1; f.ldw2 = o.ldH;
  f.Wh[0] = f.Wh[1] = params + o.w3; f.bh[0] = f.bh[1] = params + o.b3;
  f.heads = 1; f.NH = 1;
  f.h1 = h1; f.h2 = h2; f.ldh = HP; f.out[0] = f.out[1] = q; f.ldo = 1;
  f.act[0] = f.act[1] = ACT_NONE;
  f.B = B; f.H = s.H; f.split = 1 << 30;
  f.stride_params = o.count; f.stride_hidden = (int64_t)Bp * HP; f.stride_out = Bp;
  *launch_nets = nets;
  if (params2 != nullptr) {          // a second set of `nets` critics on a second input
    f.split = nets;
    f.second_params = (params2 - params) - (int64_t)nets * o.count;
    f.X2 = X2;
    *launch_nets = 2 * nets;
  }
  if (img != nullptr && img->block != nullptr) {
    f.img = FwdImages{img->block, img->v.f1, img->v.f2, {ImgView{}, ImgView{}}, img->v.bytes, 0};
    if (params2 != nullptr) f.img.second = (img->second - img->block) - (int64_t)nets * img->v.bytes;
  }
  return f;
}

int critics_forward(const float* params, CriticShape s, int nets, const float* X, int ldx, int B,
                    int Bp, float* h1, float* h2, float* q, hipStream_t st,
                    const float* params2 = nullptr, const float* X2 = nullptr,
                    const CriticImg* img = nullptr) {
  const CriticOffsets o(s);
  const int in = s.O + s.A;
  const int HP = s.hp(), H2 = s.h2();
  const int64_t hs = (int64_t)Bp * HP;
  if (s.plain() && mlp_forward_supported(s.H, 1, 1)) {              // one launch for all `nets` critics
    int launch_nets = 0;
    MlpFwdArgs f = critics_forward_args(params, s, nets, X, ldx, B, Bp, h1, h2, q, params2,
                                        X2, &launch_nets, img);
    if (params2 != nullptr) f.hidden_from = nets;   // (the first set = the targets: forward only)
    return launch_mlp_forward(f, launch_nets, st);
  }
  if (params2 != nullptr) {            // unfused path: one pass per parameter set
    TRY(critics_forward(params, s, nets, X, ldx, B, Bp, h1, h2, q, st));
    return critics_forward(params2, s, nets, X2, ldx, B, Bp, h1 + nets * hs, h2 + nets * hs,
                           q + (int64_t)nets * Bp, st);
  }
  GemmArgs g = gemm(X, ldx, params + o.W1, o.ld1, h1, HP, B, s.H, in);
  g.bias = params + o.b1; g.act = s.act;
  g.strideB = o.count; g.strideBias = o.count; g.strideC = hs;
  TRY(launch_gemm('c', 'c', g, nets, st));
  g = gemm(h1, HP, params + o.W2, o.ldH, h2, HP, B, H2, s.H);
  g.bias = params + o.b2; g.act = s.act;
  g.strideA = hs; g.strideB = o.count; g.strideBias = o.count; g.strideC = hs;
  TRY(launch_gemm('c', 'c', g, nets, st));
  g = gemm(h2, HP, params + o.w3, o.ldO, q, 1, B, 1, H2);
  g.bias = params + o.b3;
  g.strideA = hs; g.strideB = o.count; g.strideBias = o.count; g.strideC = Bp;
  TRY(launch_gemm('c', 'c', g, nets, st));
  return TONIC_OK;
}

// Backward of `nets` critics from dq [nets][Bp].  grads != null: weight/bias gradient SUMS into
// the flat layout (stride = critic param count).  dxa != null: the ACTION columns of the input
// gradient, [nets][Bp][pad16(A)] (all the actor step needs of dX; the caller adds the critics).
// loss != null: dq is not given but FORMED from the forward outputs by the step's loss (LOSS_TD /
// LOSS_ACTOR, mlpfwd.h) and written to `dq`, the logged sums to loss->stats — inside the backward
// launch when the one-launch chain applies, by the stand-alone loss kernel otherwise.
struct StepLoss {
  int kind;                                   // LOSS_TD | LOSS_ACTOR
  const float* rewards; const float* discounts; const float* tq; const float* logp;
  float alpha;
  const float* q; float* stats;
};

// The one-launch chain's arguments (mlp_backward_supported(H, 1, 0, dxa ? A : 0)).
MlpBwdArgs critics_chain_args(const float* params, CriticShape s, int nets, int B, int Bp,
                              const float* h1, const float* h2, float* dq, float* dh2, float* dh1,
                              float* dxa, const StepLoss* loss, const CriticImg* img = nullptr) {
  const CriticOffsets o(s);
  const int HP = weight_ld(s.H), ldxa = pad16(s.A);
  MlpBwdArgs b{};
  b.heads = 0; b.dq = dq; b.w3 = params + o.w3;
  if (loss) {
    b.loss = loss->kind; b.l_rewards = loss->rewards; b.l_discounts = loss->discounts;
    b.l_tq = loss->tq; b.l_logp = loss->logp; b.l_alpha = loss->alpha; b.l_q = loss->q;
    b.l_stats = loss->stats; b.l_nets = nets; b.l_Bp = Bp;
    b.l_tq_at = b.l_q_at = ValueLines{Bp, 16};
  }
  b.W2 = params + o.W2; b.W1 = params + o.W1; b.K1 = s.O + s.A; b.ldw1 = o.ld1; b.ldw2 = o.ldH;
  b.xa_first = s.O; b.xa_count = dxa ? s.A : 0;
  b.h1 = h1; b.h2 = h2; b.dz2 = dh2; b.dz1 = dh1; b.dxa = dxa; b.ldhid = HP;
  b.B = B; b.H = s.H;
  b.ldxa = ldxa;
  b.stride_params = o.count; b.stride_hidden = (int64_t)Bp * HP; b.stride_dq = Bp;
  b.stride_dxa = (int64_t)Bp * ldxa;
  if (img != nullptr && img->block != nullptr)
    b.img = BwdImages{img->block, img->v.t2, {ImgView{}, ImgView{}}, img->v.t1a, img->v.bytes};
  return b;
}

// The three weight gradients of `nets` critics (all contract over the batch) in ONE launch:
//   dw3[1,H] = dq^T h2, db3 = sum dq ; dW2[H,H] = dz2^T h1, db2 ; dW1[H,in] = dz1^T X, db1
int critics_weight_gradients(CriticShape s, int nets, const float* X, int ldx, int B, int Bp,
                             const float* h1, const float* h2, const float* dq, const float* dh2,
                             const float* dh1, float* grads, hipStream_t st, const AdamFold* fold,
                             const CriticImg* img = nullptr) {
  const CriticOffsets o(s);
  const int in = s.O + s.A;
  const int HP = s.hp(), H2 = s.h2();
  const int64_t hs = (int64_t)Bp * HP;
  GemmArgs w[3];
  w[0] = gemm(dq, 1, h2, HP, grads + o.w3, o.ldO, 1, H2, B);
  w[0].colsum = grads + o.b3; w[0].strideColsum = o.count;
  w[0].strideA = Bp; w[0].strideB = hs; w[0].strideC = o.count;
  w[1] = gemm(dh2, HP, h1, HP, grads + o.W2, o.ldH, H2, s.H, B);
  w[1].colsum = grads + o.b2; w[1].strideColsum = o.count;
  w[1].strideA = hs; w[1].strideB = hs; w[1].strideC = o.count;
  w[2] = gemm(dh1, HP, X, ldx, grads + o.W1, o.ld1, s.H, in, B);
  w[2].colsum = grads + o.b1; w[2].strideColsum = o.count;
  w[2].strideA = hs; w[2].strideC = o.count;
  if (fold != nullptr && img != nullptr && img->block != nullptr) {
    // the optimizer epilogue keeps the stepped tensors' weight images (and their targets') up to date
    const int64_t to_target = img->target != nullptr ? img->target - img->block : 0;
    w[1].img = ImgTarget{img->block + img->v.f2.off, img->v.f2.chunks, img->block + img->v.t2.off,
                         img->v.t2.chunks, 0, s.H, img->v.bytes, to_target};
    w[2].img = ImgTarget{img->block + img->v.f1.off, img->v.f1.chunks, img->block + img->v.t1a.off,
                         img->v.t1a.chunks, s.O, s.A, img->v.bytes, to_target};
  }
  return launch_gemm_group('s', 's', w, 3, nets, st, fold);
}

int critics_backward(const float* params, CriticShape s, int nets, const float* X, int ldx, int B,
                     int Bp, const float* h1, const float* h2, float* dq, float* dh2,
                     float* dh1, float* grads, float* dxa, hipStream_t st,
                     const StepLoss* loss = nullptr, const AdamFold* fold = nullptr,
                     const CriticImg* img = nullptr) {
  const CriticOffsets o(s);
  const bool one_launch = s.plain() && mlp_backward_supported(s.H, 1, 0, dxa ? s.A : 0);
  if (loss && !one_launch) {
    if (loss->kind == LOSS_TD) {
      hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(1024), 0, st, loss->rewards,
                         loss->discounts, loss->tq, loss->logp, loss->alpha, loss->q, dq,
                         loss->stats, B, Bp, nets);
    } else {
      hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(1024), 0, st, loss->q, loss->logp,
                         loss->alpha, nets == 2 ? 1 : 0, dq, loss->stats, B, Bp);
    }
  }
  const int HP = s.hp(), H2 = s.h2();
  const int64_t hs = (int64_t)Bp * HP;
  const int ldxa = pad16(s.A);
  GemmArgs g;
  // the input-gradient chain first ...
  if (one_launch) {                                               // ... in ONE launch
    MlpBwdArgs b = critics_chain_args(params, s, nets, B, Bp, h1, h2, dq, dh2, dh1, dxa, loss, img);
    b.skip_dz = grads == nullptr ? 1 : 0;          // (a frozen critic's chain: only its action columns are read)
    TRY(launch_mlp_backward(b, nets, st));
  } else {
    // dz2 = (dq w3) * relu'(h2)
    g = gemm(dq, 1, params + o.w3, o.ldO, dh2, HP, B, H2, 1);
    g.mask = h2; g.ldmask = HP; g.mask_act = s.act;
    g.strideA = Bp; g.strideB = o.count; g.strideC = hs; g.strideMask = hs;
    TRY(launch_gemm('c', 's', g, nets, st));
    // dz1 = (dz2 W2) * act'(h1)
    g = gemm(dh2, HP, params + o.W2, o.ldH, dh1, HP, B, s.H, H2);
    g.mask = h1; g.ldmask = HP; g.mask_act = s.act;
    g.strideA = hs; g.strideB = o.count; g.strideC = hs; g.strideMask = hs;
    TRY(launch_gemm('c', 's', g, nets, st));
    if (dxa) {     // dxa = dz1 W1[:, O : O + A]
      g = gemm(dh1, HP, params + o.W1 + s.O, o.ld1, dxa, ldxa, B, s.A, s.H);
      g.strideA = hs; g.strideB = o.count; g.strideC = (int64_t)Bp * ldxa;
      TRY(launch_gemm('c', 's', g, nets, st));
    }
  }
  if (grads)       // ... then the three weight gradients in ONE launch
    TRY(critics_weight_gradients(s, nets, X, ldx, B, Bp, h1, h2, dq, dh2, dh1, grads, st, fold, img));
  return TONIC_OK;
}

struct Workspace {
  char* base;
  int64_t used, capacity;
  float* take(int64_t floats) {
    float* p = reinterpret_cast<float*>(base + used);
    used += round_up(floats * 4, 256);
    return p;
  }
};

// The six networks' image blocks, carved from the END of a workspace's takes (so that nothing else moves).
struct ImageSet {
  ActorImg actor, target_actor;
  CriticImg critics, target_critics;       // (critics.second / .target = the targets and vice versa, as needed)
  bool on;
};
ImageSet take_images(Workspace& ws, int O, int A, int H, int heads, bool on) {
  ImageSet im{};
  im.on = on;
  if (!on) return im;
  const ActorImages av = actor_images(O, H, A, heads);
  const CriticImages cv = critic_images(O, A, H);
  char* a0 = reinterpret_cast<char*>(ws.take(round_up(av.bytes, 256) / 4));
  char* a1 = reinterpret_cast<char*>(ws.take(round_up(av.bytes, 256) / 4));
  char* c0 = reinterpret_cast<char*>(ws.take(round_up(2 * cv.bytes, 256) / 4));
  char* c1 = reinterpret_cast<char*>(ws.take(round_up(2 * cv.bytes, 256) / 4));
  im.actor = ActorImg{a0, a1, av};
  im.target_actor = ActorImg{a1, nullptr, av};
  im.critics = CriticImg{c0, c1, nullptr, cv};
  im.target_critics = CriticImg{c1, nullptr, c0, cv};
  return im;
}

int64_t offpolicy_workspace_floats(int B, int O, int A, int H) {
  const int64_t Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), HP = hidden_pitch(H);
  // actor h1,h2 + 2 heads + act + sigma + logp ; X ; critics h1,h2,q,dq,dh2,dh1 (x2) ; dX ; dloc,dspre,dah2,dah1
  return 2 * Bp * HP + 2 * Bp * ldh + 2 * Bp * A + Bp + Bp * ldx + 2 * (4 * Bp * HP + 2 * Bp) +
         Bp * ldx + 2 * Bp * ldh + 2 * Bp * HP + 2 * Bp * ldh + 64 * 16 +
         4 * Bp * HP + Bp * ldx + 4 * Bp +                // second input + four-network forward
         (hidden_plain(H) ? images_floats(O, A, H, 2) : 0);     // the weight images (mlpimg.h)
}

}  // namespace
}  // namespace tonic

using namespace tonic;

extern "C" int64_t tonic_offpolicy_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H) {
  return (offpolicy_workspace_floats(B, O, A, H) + 64 * 30) * 4;
}

extern "C" int32_t tonic_mlp_weight_stride(int32_t cols) { return weight_ld(cols); }

extern "C" int64_t tonic_mlp_actor_param_count(int32_t O, int32_t H, int32_t A, int32_t heads) {
  return actor_count(actor_shape(O, H, A, heads));
}
extern "C" int64_t tonic_q_critic_param_count(int32_t O, int32_t A, int32_t H) {
  return critic_count(critic_shape(O, A, H));
}

// The `H` argument of the off-policy entries for a torso other than two ReLU layers of one width:
// MLP((H1, H2), activation) with activation 1 = ReLU, 2 = Tanh, 3 = ELU (GemmAct).  -1: not representable.
extern "C" int32_t tonic_mlp_hidden(int32_t H1, int32_t H2, int32_t activation) {
  if (H1 < 1 || H1 > 4095 || H2 < 1 || H2 > 4095 || activation < ACT_RELU || activation > ACT_ELU) return -1;
  if (H1 == H2 && activation == ACT_RELU) return H1;
  return kHiddenPacked | H1 | (H2 << 12) | (activation << 24);
}

// Policy forward for acting / evaluation.  kind: 0 = deterministic tanh head (TD3,
// actors.py:113-115), 1 = squashed Gaussian sample tanh(loc + sigma * eps) (SAC
// `_stochastic_actions`, sac.py:40-43; eps = NULL gives the greedy `loc` of sac.py:48-51).
extern "C" int tonic_policy_forward(const float* d_actor_params, const float* d_observations,
                                    const float* d_eps, float* d_actions, int32_t kind, int32_t B,
                                    int32_t O, int32_t H, int32_t A, void* d_workspace,
                                    int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && d_workspace && B > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_policy_forward: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_offpolicy_workspace_bytes(B, O, A, H),
                TONIC_ERR_WORKSPACE, "tonic_policy_forward: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldh = pad16(A), HP = hidden_pitch(H);
  Workspace ws{static_cast<char*>(d_workspace), 0, workspace_bytes};
  float* h1 = ws.take((int64_t)Bp * HP); float* h2 = ws.take((int64_t)Bp * HP);
  float* head0 = ws.take((int64_t)Bp * ldh); float* head1 = ws.take((int64_t)Bp * ldh);
  const ActorShape s = actor_shape(O, H, A, kind == 0 ? 1 : 2);
  const int threads = 256;
  if (kind == 2) {       // Gaussian head with a tanh loc (MPO, mpo.py:77-85): a = loc + sigma * eps
    TRY(actor_forward(d_actor_params, s, d_observations, B, h1, h2, head0, head1, ldh, true, st));
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, head0, head1, d_eps, ldh, d_actions, B, A);
    TONIC_CHECK_LAUNCH("tonic_policy_forward");
    return TONIC_OK;
  }
  PolicyTail tail{};
  tail.post = kind == 0 ? POST_COPY : POST_SQUASHED_SAMPLE; tail.eps = d_eps; tail.actions = d_actions;
  bool tail_done = false;
  TRY(actor_forward(d_actor_params, s, d_observations, B, h1, h2, head0, head1, ldh, kind == 0, st,
                    &tail, &tail_done));
  if (tail_done) {
  } else if (kind == 0) {
    hipLaunchKernelGGL(copy_actions_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, head0, ldh, d_actions, B, A);
  } else {
    hipLaunchKernelGGL(sac_sample_kernel,
                       dim3((B * sample_group(A) + threads - 1) / threads), dim3(threads), 0, st,
                       head0, head1, d_eps, ldh, d_actions, (float*)nullptr, (float*)nullptr, B, A,
                       sample_group(A));
  }
  TONIC_CHECK_LAUNCH("tonic_policy_forward");
  return TONIC_OK;
}

// Acting on a collector's block (the off-policy agents' step when the environment writes its observations into a
// shared pinned block, tonic_amd/collector.py): ONE launch reads the W observation rows in place (page-locked host
// memory, no staging copy), runs the policy of tonic_policy_forward (kind 0 deterministic tanh head, 1 squashed
// Gaussian: eps_slot 0 = the block's first noise field holds the standard-normal draws, -1 = the greedy loc), writes
// the actions into the block's second noise field (host-visible) and one completion word per 16 rows at system
// scope; the host waits with tonic_collector_wait_actions — no event, no copy back.  d_rows_out [W, O] (may be
// NULL): a device copy of the observation rows for the transition the agent stores after the environment's step
// (the block's own rows are overwritten by then).  replaces: tonic/torch/agents/ddpg.py:45-52,
// sac.py:40-51 (`_policy` / `_greedy_actions`) for observations that live in a collector block.
extern "C" int64_t tonic_mlp_actor_image_bytes(int32_t O, int32_t H, int32_t A, int32_t heads) {
  if (!hidden_plain(H) || heads < 1 || heads > 2 || !images_serve(O, A, H) || !mlp_forward_supported(H, A, heads))
    return 0;
  return round_up(actor_images(O, H, A, heads).bytes, 256);
}

extern "C" int tonic_collector_q_act(tonic_collector_t* collector, const float* d_actor_params,
                                     void* d_actor_images, int32_t rebuild_images, int32_t kind,
                                     int32_t H, int32_t eps_slot, float* d_rows_out,
                                     const tonic_q_store_t* store, void* d_workspace,
                                     int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(collector && d_actor_params && d_actor_images && d_workspace && (kind == 0 || kind == 1),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_q_act: bad argument");
  TONIC_REQUIRE(kind == 1 || eps_slot < 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_q_act: the deterministic policy takes no noise");
  int64_t W; int O, A;
  collector_shape(collector, &W, &O, &A);      // (shapes first: nothing is opened before the arguments are known good)
  const ActorShape s = actor_shape(O, H, A, kind == 0 ? 1 : 2);
  TONIC_REQUIRE(tonic_mlp_actor_image_bytes(O, H, A, s.heads) > 0 && mlp_policy_tail_supported(s.H, s.A) &&
                    g_policy_tail != 0 && mlp_image_pass_supported(O, s.H),
                TONIC_ERR_UNSUPPORTED_SHAPE, "tonic_collector_q_act: O=%d H=%d A=%d outside the fused forward", O, H, A);
  TONIC_REQUIRE(workspace_bytes >= tonic_offpolicy_workspace_bytes((int32_t)W, O, A, H), TONIC_ERR_WORKSPACE,
                "tonic_collector_q_act: workspace too small");
  hipStream_t st = as_stream(stream);
  const int B = (int)W, Bp = pad16(B), ldh = pad16(A), HP = hidden_pitch(H);
  Workspace ws{static_cast<char*>(d_workspace), 0, workspace_bytes};
  float* h1 = ws.take((int64_t)Bp * HP); float* h2 = ws.take((int64_t)Bp * HP);
  float* head0 = ws.take((int64_t)Bp * ldh); float* head1 = ws.take((int64_t)Bp * ldh);
  // the actor's weight images (mlpimg.h): the caller's buffer, rebuilt here when it says the parameters moved
  const ActorImg img{static_cast<char*>(d_actor_images), nullptr, actor_images(O, s.H, A, s.heads)};
  if (rebuild_images != 0) {
    ImgBuild build;
    add_actor_images(build, d_actor_params, s, img);
    TRY(launch_build_images(build, st));
  }
  TONIC_REQUIRE(store == nullptr ||
                    (store->d_buf_observations && store->d_buf_actions && store->d_buf_next_observations &&
                     store->d_buf_rewards && store->d_buf_resets && store->d_buf_terminations &&
                     store->d_buf_discounts && store->d_observations && store->row >= 0),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_q_act: bad store argument");
  CollectorStep step{};
  TRY(collector_begin_q_step(collector, eps_slot, store != nullptr ? 1 : 0, &step));
  BufferStoreArgs st_args{};
  if (store != nullptr) {      // the previous step's transition: the block's outcome fields, read in place
    st_args = BufferStoreArgs{store->d_buf_observations, store->d_buf_actions, store->d_buf_next_observations,
                              store->d_buf_rewards, store->d_buf_resets, store->d_buf_terminations,
                              store->d_buf_discounts, store->d_observations, step.actions, step.next_observations,
                              step.rewards, step.resets, step.terminations, store->d_norm_acc, store->row, W, O, A,
                              (float)store->discount_factor};
  }
  PolicyTail tail{};
  tail.post = kind == 0 ? POST_COPY : POST_SQUASHED_SAMPLE; tail.eps = step.eps; tail.actions = step.actions_out;
  bool tail_done = false;
  TRY(actor_forward(d_actor_params, s, step.observations, B, h1, h2, head0, head1, ldh, kind == 0, st, &tail,
                    &tail_done, 0, &img, &step, d_rows_out, store != nullptr ? &st_args : nullptr));
  TONIC_REQUIRE(tail_done, TONIC_ERR_UNSUPPORTED_SHAPE, "tonic_collector_q_act: the policy tail did not fuse");
  TONIC_CHECK_LAUNCH("tonic_collector_q_act");
  return TONIC_OK;
}

// Q-learning gradients.  kind 0 = TD3 (critics.py:156-175: target actor + clipped noise),
// 1 = SAC (critics.py:202-227: online actor sample, entropy term), 2 = DDPG (critics.py:68-86:
// ONE critic, target actor, no noise; gradient sums for [critic] + 8 statistics).  Writes gradient SUMS
// for [critic_1 | critic_2] + 8 statistics {sq_err_sum(both), q1_sum, q2_sum, 0, 0, B, 0, 0}
// into d_grad_sums; the caller follows with tonic_adam_step(grad_scale = 1/B).
extern "C" int tonic_twin_q_grad(int32_t kind, const float* d_policy_params,
                                 const float* d_target_critics, const float* d_critics,
                                 const float* d_norm_mean, const float* d_norm_std,
                                 double norm_clip,
                                 const float* d_observations, const float* d_actions,
                                 const float* d_next_observations, const float* d_rewards,
                                 const float* d_discounts, const float* d_eps,
                                 float* d_grad_sums, int32_t B, int32_t O, int32_t H, int32_t A,
                                 double entropy_coeff, double noise_scale, double noise_clip,
                                 void* d_workspace, int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_policy_params && d_target_critics && d_critics && d_norm_mean && d_norm_std &&
                    d_observations && d_actions && d_next_observations && d_rewards &&
                    d_discounts && (d_eps || kind == 2) && d_grad_sums && d_workspace && B > 0 &&
                    kind >= 0 && kind <= 2,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_twin_q_grad: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_offpolicy_workspace_bytes(B, O, A, H),
                TONIC_ERR_WORKSPACE, "tonic_twin_q_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), threads = 256, HP = hidden_pitch(H);
  const CriticShape cs = critic_shape(O, A, H);
  const int64_t Pc = critic_count(cs);
  Workspace ws{static_cast<char*>(d_workspace), 0, workspace_bytes};
  float* a_h1 = ws.take((int64_t)Bp * HP); float* a_h2 = ws.take((int64_t)Bp * HP);
  float* head0 = ws.take((int64_t)Bp * ldh); float* head1 = ws.take((int64_t)Bp * ldh);
  float* next_act = ws.take((int64_t)Bp * A); float* logp = ws.take(Bp);
  const int nets = kind == 2 ? 1 : 2;
  const int64_t hs = (int64_t)Bp * HP;
  // target critics on (s', a') and online critics on (s, a) share ONE forward launch: inputs X /
  // X2, activations and values laid out [targets | online]
  float* X = ws.take((int64_t)Bp * ldx); float* X2 = ws.take((int64_t)Bp * ldx);
  float* h1_all = ws.take(4 * hs); float* h2_all = ws.take(4 * hs);
  float* q_all = ws.take(4LL * Bp);
  float* c_h1 = h1_all + nets * hs; float* c_h2 = h2_all + nets * hs;
  float* tq = q_all; float* q = q_all + (int64_t)nets * Bp;
  float* dq = ws.take(2LL * Bp);
  float* dh2 = ws.take(2 * hs); float* dh1 = ws.take(2 * hs);

  // ---- targets (no grad)
  const ActorShape as = actor_shape(O, H, A, kind == 1 ? 2 : 1);
  // the fused passes' weight images (mlpimg.h), formed from the float32 parameters by THIS call: the policy's,
  // the target critics' and the critics' — the same conversion as the fused iteration's, so the same bits
  ImageSet im = take_images(ws, O, A, H, as.heads, hidden_plain(H) && images_serve(O, A, H) &&
                                                    mlp_forward_supported(H, A, as.heads) &&
                                                    mlp_backward_supported(H, 1, 0, 0));
  if (im.on) {
    ImgBuild build;
    add_actor_images(build, d_policy_params, as, im.actor);
    add_critic_images(build, d_critics, cs, nets, im.critics.block, im.critics.v);
    add_critic_images(build, d_target_critics, cs, nets, im.target_critics.block, im.target_critics.v);
    TRY(launch_build_images(build, st));
  }
  // the policy's tail also encodes both critic inputs: (s', a') from its own actions -> X, the
  // stored (s, a) -> X2
  PolicyTail tail{};
  tail.post = kind == 0 ? POST_TARGET_NOISE : kind == 2 ? POST_COPY : POST_SQUASHED_SAMPLE;
  tail.eps = d_eps; tail.actions = next_act; tail.logp = kind == 1 ? logp : nullptr;
  tail.noise_scale = (float)noise_scale; tail.noise_clip = (float)noise_clip;
  tail.enc_obs = d_next_observations; tail.enc_obs2 = d_observations; tail.enc_act2 = d_actions;
  tail.enc_mean = d_norm_mean; tail.enc_std = d_norm_std; tail.enc_clip = clip_bound(norm_clip);
  tail.enc_out = X; tail.enc_out2 = X2; tail.enc_ld = ldx;
  bool tail_done = false;
  TRY(actor_forward(d_policy_params, as, d_next_observations, B, a_h1, a_h2, head0, head1, ldh,
                    kind != 1, st, &tail, &tail_done, 0, im.on ? &im.actor : nullptr));
  if (tail_done) {
  } else if (kind == 0) {
    hipLaunchKernelGGL(td3_target_action_kernel, dim3((B * A + threads - 1) / threads),
                       dim3(threads), 0, st, head0, ldh, d_eps, next_act, B, A,
                       (float)noise_scale, (float)noise_clip);
  } else if (kind == 2) {
    hipLaunchKernelGGL(copy_actions_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, head0, ldh, next_act, B, A);
  } else {
    hipLaunchKernelGGL(sac_sample_kernel,
                       dim3((B * sample_group(A) + threads - 1) / threads), dim3(threads), 0, st,
                       head0, head1, d_eps, ldh, next_act, logp, (float*)nullptr, B, A,
                       sample_group(A));
  }
  // ---- inputs of the targets (s', a') and of the online critics (s, a): one launch each for
  //      the encoding and for the four-network forward
  if (!tail_done) {
    hipLaunchKernelGGL(encode_kernel, dim3((B * (O + A) + threads - 1) / threads, 2),
                       dim3(threads), 0, st, d_next_observations, next_act, d_norm_mean, d_norm_std,
                       clip_bound(norm_clip), X, B, O, A, ldx, d_observations, d_actions, X2);
  }
  TRY(critics_forward(d_target_critics, cs, nets, X, ldx, B, Bp, h1_all, h2_all, q_all, st,
                      d_critics, X2, im.on ? &im.target_critics : nullptr));
  const StepLoss td{LOSS_TD, d_rewards, d_discounts, tq, kind == 1 ? logp : (const float*)nullptr,
                    (float)entropy_coeff, q, d_grad_sums + nets * Pc};
  TRY(critics_backward(d_critics, cs, nets, X2, ldx, B, Bp, c_h1, c_h2, dq, dh2, dh1, d_grad_sums,
                       nullptr, st, &td, nullptr, im.on ? &im.critics : nullptr));
  TONIC_CHECK_LAUNCH("tonic_twin_q_grad");
  return TONIC_OK;
}

namespace {

// Backward of an actor-shaped network (torso + `heads` linear heads) from the gradients at its head
// outputs: the input-gradient chain (dz2, dz1; optionally the columns [xa_first, xa_first + xa_count)
// of the input gradient -> dxa), then all weight / bias gradient SUMS into the flat layout `grads`
// (null: none — a frozen network) in one grouped launch.  X: the network's input rows, ldx apart.
MlpBwdArgs actor_chain_args(const float* params, ActorShape as, int B, const float* a_h1,
                            const float* a_h2, const float* dloc, const float* dspre, int ldh,
                            float* da_h2, float* da_h1, float* dxa, int xa_first, int xa_count,
                            const MlpBwdArgs* head_fold, const ActorImg* img = nullptr) {
  const int H = as.H, A = as.A, HP = as.hp();
  ActorParams p(params, as);
  MlpBwdArgs b{};
  b.heads = as.heads; b.NH = A; b.ldh = ldh;
  b.dhead[0] = dloc; b.dhead[1] = dspre ? dspre : dloc;
  b.Wh[0] = p.head_w(0); b.Wh[1] = p.head_w(as.heads - 1);
  b.W2 = p.W2; b.W1 = p.W1; b.K1 = as.O; b.ldw1 = p.ld1; b.ldw2 = p.ldH;
  b.xa_first = xa_first; b.xa_count = dxa ? xa_count : 0;
  b.h1 = a_h1; b.h2 = a_h2; b.dz2 = da_h2; b.dz1 = da_h1; b.dxa = dxa; b.ldhid = HP;
  b.ldxa = pad16(xa_count > 0 ? xa_count : 1);
  b.B = B; b.H = H;
  if (head_fold != nullptr) {          // dloc / dspre are FORMED by this launch (hb_* of MlpBwdArgs)
    b.hb_dxa0 = head_fold->hb_dxa0; b.hb_dxa1 = head_fold->hb_dxa1; b.hb_ldxa = head_fold->hb_ldxa;
    b.hb_act = head_fold->hb_act; b.hb_eps = head_fold->hb_eps; b.hb_sigma = head_fold->hb_sigma;
    b.hb_spre = head_fold->hb_spre; b.hb_sac = head_fold->hb_sac; b.hb_alpha = head_fold->hb_alpha;
  }
  if (img != nullptr && img->block != nullptr && b.xa_count == 0)     // (no image of W1^T's columns for actors)
    b.img = BwdImages{img->block, img->v.t2, {img->v.th[0], img->v.th[as.heads - 1]}, ImgView{}, 0};
  return b;
}

// All weight gradients of an actor-shaped network (they contract over the batch) in ONE launch:
//   dWh[A,H] = dhead^T h2, dbh (per head) ; dW2 = dz2^T h1, db2 ; dW1 = dz1^T X, db1
int actor_weight_gradients(ActorShape as, const float* X, int ldx, int B, const float* a_h1,
                           const float* a_h2, const float* dloc, const float* dspre, int ldh,
                           const float* da_h2, const float* da_h1, float* grads, hipStream_t st,
                           const AdamFold* fold, const ActorImg* img = nullptr) {
  const int H = as.H, A = as.A, HP = as.hp(), H2 = as.h2();
  const ActorBlock<float> gp(grads, as);               // the gradient sums share the layout
  GemmArgs w[4];
  int count = 0;
  const bool images = fold != nullptr && img != nullptr && img->block != nullptr;
  const int64_t to_target = images && img->target != nullptr ? img->target - img->block : 0;
  for (int h = 0; h < as.heads; ++h) {
    const float* dhead = h == 0 ? dloc : dspre;
    w[count] = gemm(dhead, ldh, a_h2, HP, gp.head_w(h), gp.ldO, A, H2, B);
    if (images)
      w[count].img = ImgTarget{img->block + img->v.fh[h].off, img->v.fh[h].chunks, img->block + img->v.th[h].off,
                               img->v.th[h].chunks, 0, H2, 0, to_target};
    w[count++].colsum = gp.head_b(h);
  }
  w[count] = gemm(da_h2, HP, a_h1, HP, gp.W2, gp.ldH, H2, H, B);
  if (images)
    w[count].img = ImgTarget{img->block + img->v.f2.off, img->v.f2.chunks, img->block + img->v.t2.off,
                             img->v.t2.chunks, 0, H, 0, to_target};
  w[count++].colsum = gp.b2;
  w[count] = gemm(da_h1, HP, X, ldx, gp.W1, gp.ld1, H, as.O, B);
  if (images) w[count].img = ImgTarget{img->block + img->v.f1.off, img->v.f1.chunks, nullptr, 0, 0, 0, 0, to_target};
  w[count++].colsum = gp.b1;
  return launch_gemm_group('s', 's', w, count, 1, st, fold);
}

int actor_shaped_backward(const float* params, ActorShape as, const float* X, int ldx, int B,
                          const float* a_h1, const float* a_h2, const float* dloc,
                          const float* dspre, int ldh, float* da_h2, float* da_h1, float* grads,
                          float* dxa, int xa_first, int xa_count, hipStream_t st,
                          const MlpBwdArgs* head_fold = nullptr, const AdamFold* fold = nullptr,
                          const ActorImg* img = nullptr) {
  const int H = as.H, A = as.A, HP = as.hp(), H2 = as.h2();
  ActorParams p(params, as);
  GemmArgs g;
  // the input-gradient chain first: dz2 = (dloc Wloc [+ dspre Wscale]) * act'(h2) ; dz1
  if (as.plain() && mlp_backward_supported(H, A, as.heads, xa_count)) {
    const MlpBwdArgs b = actor_chain_args(params, as, B, a_h1, a_h2, dloc, dspre, ldh, da_h2, da_h1,
                                          dxa, xa_first, xa_count, head_fold, img);
    TRY(launch_mlp_backward(b, 1, st));
  } else {
    for (int h = 0; h < as.heads; ++h) {
      const float* dhead = h == 0 ? dloc : dspre;
      g = gemm(dhead, ldh, p.head_w(h), p.ldO, da_h2, HP, B, H2, A);
      g.mask = a_h2; g.ldmask = HP; g.accumulate = h > 0; g.mask_act = as.act;
      TRY(launch_gemm('c', 's', g, 1, st));
    }
    g = gemm(da_h2, HP, p.W2, p.ldH, da_h1, HP, B, H, H2);
    g.mask = a_h1; g.ldmask = HP; g.mask_act = as.act;
    TRY(launch_gemm('c', 's', g, 1, st));
    if (dxa) {
      g = gemm(da_h1, HP, p.W1 + xa_first, p.ld1, dxa, pad16(xa_count), B, xa_count, H);
      TRY(launch_gemm('c', 's', g, 1, st));
    }
  }
  if (grads == nullptr) return TONIC_OK;
  return actor_weight_gradients(as, X, ldx, B, a_h1, a_h2, dloc, dspre, ldh, da_h2, da_h1, grads, st,
                                fold, img);
}

}  // namespace

// Actor gradient through the (frozen) critics.  kind 0 = DeterministicPolicyGradient on
// critic_1 only (actors.py:170-189 with td3.py:36), 1 = TwinCriticSoftDeterministicPolicyGradient

// ------------------------------------------------------------------ one whole learner iteration
// ddpg.py:95-112 / td3.py:38-55 / sac.py with the launches of the two updaters merged where they do
// not depend on each other, and the optimizer steps folded into the weight-gradient launches:
//   1  policy passes of BOTH steps: net 0 = the critic step's policy on s' (TD3 / DDPG: target actor
//      [+ clipped noise], SAC: online actor sample + log-prob) whose tail also encodes the critics'
//      inputs (s', a') and (s, a); net 1 (actor due) = the online actor on s with the actor step's
//      sample, whose tail encodes (s, a_new).  The critic step does not touch the actor, so net 1
//      sees the parameters the reference's actor step would see.
//   2  target critics on (s', a') + online critics on (s, a)
//   3  TD loss + the online critics' input-gradient chain
//   4  the critics' weight gradients + Adam [+ polyak of the critics when the targets move]
//   5  (actor due) the UPDATED critics on (s, a_new)
//   6  actor objective + the critics' chain down to the action columns
//   7  head backward + the actor's input-gradient chain
//   8  the actor's weight gradients + Adam + polyak of the actor
// 13 launches of the split path (gather, 4 forwards, 3 backwards, head backward, 2 weight-gradient
// groups, 2 optimizer launches) become 8 (4 when the actor is not due).
namespace {

// The one word of the chained launches that must be ZERO when a launch starts (a reader sets it when
// a value never came, the launch's last workgroup reports and clears it) lives at the START of the
// workspace, so that it stays where it is whatever batch size the workspace is used with next.
constexpr int64_t kChainSyncArea = 64;               // floats

int64_t q_iteration_floats(int B, int O, int A, int H) {
  const int64_t Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), HP = weight_ld(H);
  // what the policy passes (launch 1) write and the chained launches exchange: two sets (tonic_q_iteration_t.slot)
  const int64_t set = 2 * (2 * Bp * HP + 2 * Bp * ldh)   // policy passes: h1, h2, two head outputs, x2
                      + 3 * Bp * A + 2 * Bp              // next actions, new actions, sigma, two log-probs
                      + 3 * Bp * ldx                     // X (s', a'), X2 (s, a), X3 (s, a_new)
                      + 2 * Bp * ldh                     // dxa (two critics)
                      + (Bp / 16) * kExchangeTileFloats; // the chained launches' value lines
  return 2 * set
         + 2 * 4 * Bp * HP + 4 * Bp                 // four-critic forward: h1, h2, values
         + 2 * Bp + 2 * 2 * Bp * HP                 // dq, dz2, dz1 of two critics
         + 2 * Bp * ldh                             // dloc, dspre
         + 2 * Bp * HP                              // actor dz2, dz1
         + kChainSyncArea                           // the chained launches' failure words (first)
         + images_floats(O, A, H, 2);               // the six networks' weight images (mlpimg.h)
}

}  // namespace

extern "C" int64_t tonic_q_iteration_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H) {
  return (q_iteration_floats(B, O, A, H) + 64 * 64) * 4;       // (+ the 256-byte rounding of every take)
}

extern "C" int tonic_q_iteration_ahead_supported(int32_t B, int32_t O, int32_t H, int32_t A, int32_t nets,
                                                 int32_t passes) {
  if (B <= 0 || nets < 1 || nets > 2 || passes < 1 || passes > 2) return 0;
  if (g_q_chain.load() == 0 || !hidden_plain(H) || !images_serve(O, A, H) || !mlp_image_pass_supported(O, H)) return 0;
  const int tiles = (B + 15) / 16;
  return tiles * (2 * nets + passes) + 1 <= 256;      // one workgroup per compute unit: all of them at once
}

extern "C" int tonic_q_iteration_supported(int32_t O, int32_t H, int32_t A, int32_t heads) {
  return hidden_plain(H) && mlp_forward_supported(H, A, heads) && mlp_policy_tail_supported(H, A) &&
         mlp_forward_supported(H, 1, 1) && mlp_backward_supported(H, 1, 0, A) &&
         mlp_backward_supported(H, A, heads, 0) && O > 0 ? 1 : 0;
}

extern "C" int tonic_q_iteration(const tonic_q_iteration_t* it, void* stream) {
  TONIC_REQUIRE(it != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_q_iteration: null arguments");
  const tonic_q_iteration_t& a = *it;
  const int kind = a.kind, B = a.B, O = a.O, H = a.H, A = a.A;
  const bool due = a.actor_due != 0;
  const int phase = a.phase;                 // 0 whole iteration | 1 critic half | 2 actor half (sums only)
  TONIC_REQUIRE(phase >= 0 && phase <= 2 && (phase != 2 || due), TONIC_ERR_INVALID_ARGUMENT,
                "tonic_q_iteration: phase %d (actor_due %d)", phase, (int)due);
  const int stage = a.stage, slot = a.slot;  // 0 whole | 2 behind the policy passes; the set of launch-1 outputs
  TONIC_REQUIRE((stage == 0 || stage == 2) && (stage == 0 || phase == 0) && (slot == 0 || slot == 1),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_q_iteration: stage %d, slot %d (phase %d)", stage, slot, phase);
  const tonic_q_iteration_t* next = a.ahead;           // the iteration whose policy passes ride in this critic step
  TONIC_REQUIRE(next == nullptr ||
                    (phase == 0 && !due && next->kind == kind && next->B == B && next->O == O && next->H == H &&
                     next->A == A && next->slot == (slot ^ 1) && next->d_workspace == a.d_workspace &&
                     next->d_actor == a.d_actor && next->d_target_actor == a.d_target_actor &&
                     next->d_next_observations && next->d_observations && next->d_actions &&
                     (next->d_eps_critic || kind == 2) && (next->d_eps_actor || kind != 1 || !next->actor_due) &&
                     tonic_q_iteration_ahead_supported(B, O, H, A, kind == 2 ? 1 : 2, next->actor_due ? 2 : 1)),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_q_iteration: policy passes ahead (actor_due %d)", (int)due);
  TONIC_REQUIRE(kind >= 0 && kind <= 2 && B > 0 && a.d_actor && a.d_critics && a.d_target_actor &&
                    a.d_target_critics && a.d_norm_mean && a.d_norm_std && a.d_observations &&
                    a.d_actions && a.d_next_observations && a.d_rewards && a.d_discounts &&
                    (a.d_eps_critic || kind == 2) && (a.d_eps_actor || kind != 1 || !due) &&
                    a.critic.d_grad_sums && a.critic.d_exp_avg && a.critic.d_exp_avg_sq &&
                    a.critic.d_state && a.actor.d_grad_sums && a.actor.d_exp_avg &&
                    a.actor.d_exp_avg_sq && a.actor.d_state && a.d_workspace,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_q_iteration: bad argument");
  const int heads = kind == 1 ? 2 : 1;
  TONIC_REQUIRE(tonic_q_iteration_supported(O, H, A, heads), TONIC_ERR_UNSUPPORTED_SHAPE,
                "tonic_q_iteration: O=%d H=%d A=%d outside the fused kernels", O, H, A);
  TONIC_REQUIRE(a.workspace_bytes >= tonic_q_iteration_workspace_bytes(B, O, A, H),
                TONIC_ERR_WORKSPACE, "tonic_q_iteration: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), HP = weight_ld(H);
  const int nets = kind == 2 ? 1 : 2;                  // critics
  const CriticShape cs{O, A, H};
  const ActorShape as{O, H, A, heads};
  const int64_t Pc = critic_count(cs), Pa = actor_count(as), hs = (int64_t)Bp * HP;
  Workspace ws{static_cast<char*>(a.d_workspace), 0, a.workspace_bytes};
  // (one failure word per set: the passes that run ahead clear THEIR iteration's word while this one's launches may set theirs)
  unsigned* failed_words = reinterpret_cast<unsigned*>(ws.take(kChainSyncArea));
  unsigned* failed = failed_words + slot;
  // Everything launch 1 writes — the chained launches' exchange area (the value lines of the tiles, then the
  // critics' action-column gradients: ONE span, emptied by the policy launch of every iteration), the policy
  // passes' activations and head outputs [net 0 | net 1], the actions / log-probabilities, the critics' input
  // rows — twice: an iteration works on set `slot`, the passes it runs ahead for the next one on the other
  struct PolicySet {
    float *xq, *dxa, *exchange_end, *p_h1, *p_h2, *head0, *head1, *next_act, *logp_next, *act, *sigma, *logp,
        *X, *X2, *X3;
  } sets[2];
  for (int set = 0; set < 2; ++set) {
    PolicySet& t = sets[set];
    t.xq = ws.take((int64_t)(Bp / 16) * kExchangeTileFloats);
    t.dxa = ws.take(2LL * Bp * ldh);
    t.exchange_end = ws.take(0);
    t.p_h1 = ws.take(2 * hs); t.p_h2 = ws.take(2 * hs);
    t.head0 = ws.take(2LL * Bp * ldh); t.head1 = ws.take(2LL * Bp * ldh);
    t.next_act = ws.take((int64_t)Bp * A); t.logp_next = ws.take(Bp);
    t.act = ws.take((int64_t)Bp * A); t.sigma = ws.take((int64_t)Bp * A);
    t.logp = ws.take(Bp);
    t.X = ws.take((int64_t)Bp * ldx); t.X2 = ws.take((int64_t)Bp * ldx);
    t.X3 = ws.take((int64_t)Bp * ldx);
  }
  const PolicySet& mine = sets[slot];
  float *xq = mine.xq, *dxa = mine.dxa, *p_h1 = mine.p_h1, *p_h2 = mine.p_h2, *head1 = mine.head1,
        *logp_next = mine.logp_next, *act = mine.act, *sigma = mine.sigma, *logp = mine.logp, *X = mine.X,
        *X2 = mine.X2, *X3 = mine.X3;
  const bool chain = g_q_chain.load() != 0;
  float* h1_all = ws.take(4 * hs); float* h2_all = ws.take(4 * hs);
  float* q_all = ws.take(4LL * Bp);
  float* c_h1 = h1_all + nets * hs; float* c_h2 = h2_all + nets * hs;
  float* tq = q_all; float* q = q_all + (int64_t)nets * Bp;
  float* dq = ws.take(2LL * Bp);
  float* dh2 = ws.take(2 * hs); float* dh1 = ws.take(2 * hs);
  float* dloc = ws.take((int64_t)Bp * ldh); float* dspre = ws.take((int64_t)Bp * ldh);
  float* da_h2 = ws.take(hs); float* da_h1 = ws.take(hs);
  // the weight images of the six networks (mlpimg.h): rebuilt from the float32 parameters when the caller says
  // they may have changed since this workspace last saw them (the first iteration of an update call; every call
  // in phases, where the caller's own optimizer launches step the parameters), kept up to date by the optimizer
  // epilogues of this call otherwise
  ImageSet im = take_images(ws, O, A, H, heads, images_serve(O, A, H));
  if (im.on && (a.refresh_images != 0 || phase != 0)) {
    ImgBuild build;
    add_actor_images(build, a.d_actor, as, im.actor);
    add_actor_images(build, a.d_target_actor, as, im.target_actor);
    add_critic_images(build, a.d_critics, cs, nets, im.critics.block, im.critics.v);
    add_critic_images(build, a.d_target_critics, cs, nets, im.target_critics.block, im.target_critics.v);
    TRY(launch_build_images(build, st));
  }
  const ActorImg* actor_img = im.on ? &im.actor : nullptr;
  const CriticImg* critics_img = im.on ? &im.critics : nullptr;

  // ---- 1: the policy passes (of iteration `it` into set `t`: this one's, or the next one's ahead of time)
  auto policy_passes = [&](const tonic_q_iteration_t& it, const PolicySet& t, unsigned* failed_word) {
    const bool it_due = it.actor_due != 0;
    const float* policy = kind == 1 ? it.d_actor : it.d_target_actor;
    ActorParams p(policy, as);
    MlpFwdArgs f{};
    f.X = it.d_next_observations; f.ldx = O; f.K1 = O;
    f.W1 = p.W1; f.b1 = p.b1; f.W2 = p.W2; f.b2 = p.b2; f.ldw1 = p.ld1; f.ldw2 = p.ldH;
    f.Wh[0] = p.head_w(0); f.bh[0] = p.head_b(0);
    f.Wh[1] = p.head_w(heads - 1); f.bh[1] = p.head_b(heads - 1);
    f.heads = heads; f.NH = A;
    f.h1 = t.p_h1; f.h2 = t.p_h2; f.ldh = HP;
    f.out[0] = t.head0; f.out[1] = heads == 2 ? t.head1 : t.head0; f.ldo = ldh;
    f.act[0] = kind != 1 ? ACT_TANH : ACT_NONE; f.act[1] = ACT_NONE;
    f.B = B; f.H = H; f.split = 1 << 30;
    f.stride_hidden = hs; f.stride_out = (int64_t)Bp * ldh;
    f.post = kind == 0 ? POST_TARGET_NOISE : kind == 2 ? POST_COPY : POST_SQUASHED_SAMPLE;
    f.post_eps = it.d_eps_critic; f.post_actions = t.next_act;
    f.post_logp = kind == 1 ? t.logp_next : nullptr;
    f.noise_scale = (float)it.noise_scale; f.noise_clip = (float)it.noise_clip;
    f.enc_obs = it.d_next_observations; f.enc_obs2 = it.d_observations; f.enc_act2 = it.d_actions;
    f.enc_mean = it.d_norm_mean; f.enc_std = it.d_norm_std; f.enc_clip = clip_bound(it.norm_clip);
    f.enc_out = t.X; f.enc_out2 = t.X2; f.enc_O = O; f.enc_ld = ldx;
    // (phases: the failure word is the caller's to clear — it stands for the whole update)
    if (chain) { f.reset_area = t.xq; f.reset_floats = t.exchange_end - t.xq; f.reset_failed = phase == 0 ? failed_word : nullptr; }
    if (it_due) {
      f.split = 1;
      f.second_params = it.d_actor - policy;           // (0 for SAC: the same network on s)
      f.X2 = it.d_observations;
      f.tail2.post = kind == 1 ? POST_SQUASHED_SAMPLE : POST_COPY;
      f.tail2.eps = kind == 1 ? it.d_eps_actor : nullptr;
      f.tail2.actions = t.act; f.tail2.sigma = kind == 1 ? t.sigma : nullptr;
      f.tail2.logp = kind == 1 ? t.logp : nullptr;
      f.tail2.enc_obs = it.d_observations; f.tail2.enc_out = t.X3;
    }
    if (im.on) {
      const ActorImg& pi = kind == 1 ? im.actor : im.target_actor;
      f.img = FwdImages{pi.block, pi.v.f1, pi.v.f2, {pi.v.fh[0], pi.v.fh[heads - 1]}, 0,
                        it_due ? im.actor.block - pi.block : 0};
      f.hidden_from = 1;             // (net 0, the critic step's policy, has no backward)
    }
    return f;
  };
  if (phase != 2 && stage != 2) TRY(launch_mlp_forward(policy_passes(a, mine, failed), due ? 2 : 1, st));
  // ---- 2: targets on (s', a') and online critics on (s, a)
  // ---- 3 + 4: TD loss, backward chain, weight gradients + Adam (+ polyak of the critics)
  const float grad_scale = (float)(1.0 / (a.global_batch > 0 ? a.global_batch : B));
  AdamFold cf{};
  cf.grads = a.critic.d_grad_sums; cf.params = a.d_critics; cf.exp_avg = a.critic.d_exp_avg;
  cf.exp_avg_sq = a.critic.d_exp_avg_sq; cf.state = a.critic.d_state; cf.n = nets * Pc;
  cf.grad_scale = grad_scale; cf.beta2 = (float)a.critic.beta2; cf.eps = (float)a.critic.eps;
  cf.beta1_d = a.critic.beta1; cf.beta2_d = a.critic.beta2; cf.lr_d = a.critic.lr;
  cf.stats_kind = 3; cf.info_row = a.critic.d_info_row; cf.consts = a.critic.d_step_constants;
  cf.skip = chain ? failed : nullptr;
  if (due) {
    cf.target = a.d_target_critics;
    cf.polyak_keep = (float)(1.0 - a.target_coeff); cf.polyak_mix = (float)a.target_coeff;
  }
  const StepLoss td{LOSS_TD, a.d_rewards, a.d_discounts, tq,
                    kind == 1 ? logp_next : (const float*)nullptr, (float)a.critic_entropy_coeff, q,
                    a.critic.d_grad_sums + nets * Pc};
  const AdamFold* critic_fold = phase == 0 ? &cf : nullptr;       // phases: gradient sums only
  if (phase == 2) {
    // (the critic half ran in an earlier call)
  } else if (chain) {
    // 2 + 3 as ONE launch: the online critics' workgroups go on to the TD loss and their chain as
    // soon as the targets of their 16 rows have arrived (q_critic_step_kernel)
    QCriticStep step{};
    int launch_nets = 0;
    step.fwd = critics_forward_args(a.d_target_critics, cs, nets, X, ldx, B, Bp, h1_all, h2_all,
                                    q_all, a.d_critics, X2, &launch_nets, im.on ? &im.target_critics : nullptr);
    step.fwd.xq = xq;
    step.fwd.hidden_from = nets;     // (the targets: forward only)
    step.bwd = critics_chain_args(a.d_critics, cs, nets, B, Bp, c_h1, c_h2, dq, dh2, dh1, nullptr, &td,
                                  critics_img);
    step.bwd.exchange_failed = failed;   // targets: lines 0, 1 of the tile; the online critics: lines 2, 3
    step.bwd.l_tq = xq; step.bwd.l_tq_at = ValueLines{32, kExchangeTileFloats};
    step.bwd.l_q = xq + 64; step.bwd.l_q_at = ValueLines{32, kExchangeTileFloats};
    step.nets = nets;
    if (next != nullptr) {            // the next iteration's policy passes: more workgroups of this launch
      step.ahead = policy_passes(*next, sets[slot ^ 1], failed_words + (slot ^ 1));
      step.ahead_nets = next->actor_due ? 2 : 1;
    }
    TRY(launch_q_critic_step(step, st));
    TRY(critics_weight_gradients(cs, nets, X2, ldx, B, Bp, c_h1, c_h2, dq, dh2, dh1,
                                 a.critic.d_grad_sums, st, critic_fold, critics_img));
  } else {
    TONIC_REQUIRE(next == nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_q_iteration: passes ahead need the chained launches");
    TRY(critics_forward(a.d_target_critics, cs, nets, X, ldx, B, Bp, h1_all, h2_all, q_all, st,
                        a.d_critics, X2, im.on ? &im.target_critics : nullptr));
    TRY(critics_backward(a.d_critics, cs, nets, X2, ldx, B, Bp, c_h1, c_h2, dq, dh2, dh1,
                         a.critic.d_grad_sums, nullptr, st, &td, critic_fold, critics_img));
  }
  if (!due || phase == 1) {
    TONIC_CHECK_LAUNCH("tonic_q_iteration");
    return TONIC_OK;
  }
  // ---- 5 + 6: the updated critics on (s, a_new), the actor objective, down to the action columns
  const int used = kind == 1 ? 2 : 1;                  // TD3 / DDPG: critic_1 only (td3.py:36)
  const StepLoss objective{LOSS_ACTOR, nullptr, nullptr, nullptr, logp,
                           (float)a.actor_entropy_coeff, q, a.actor.d_grad_sums + Pa};
  // ---- 7 + 8: head backward + actor chain, weight gradients + Adam + polyak of the actor
  MlpBwdArgs hb{};
  hb.hb_dxa0 = dxa; hb.hb_dxa1 = used == 2 ? dxa + (int64_t)Bp * ldh : nullptr; hb.hb_ldxa = ldh;
  hb.hb_act = act; hb.hb_eps = a.d_eps_actor; hb.hb_sigma = sigma;
  hb.hb_spre = head1 + (int64_t)Bp * ldh;              // net 1's scale head
  hb.hb_sac = kind == 1 ? 1 : 0; hb.hb_alpha = (float)a.actor_entropy_coeff;
  AdamFold af{};
  af.grads = a.actor.d_grad_sums; af.params = a.d_actor; af.exp_avg = a.actor.d_exp_avg;
  af.exp_avg_sq = a.actor.d_exp_avg_sq; af.state = a.actor.d_state; af.n = Pa;
  af.grad_scale = grad_scale; af.beta2 = (float)a.actor.beta2; af.eps = (float)a.actor.eps;
  af.beta1_d = a.actor.beta1; af.beta2_d = a.actor.beta2; af.lr_d = a.actor.lr;
  af.stats_kind = 4; af.info_row = a.actor.d_info_row; af.consts = a.actor.d_step_constants;
  af.skip = chain ? failed : nullptr;
  af.target = a.d_target_actor;
  af.polyak_keep = (float)(1.0 - a.target_coeff); af.polyak_mix = (float)a.target_coeff;
  const AdamFold* actor_fold = phase == 0 ? &af : nullptr;
  if (chain) {
    // 5 + 6 + 7 as ONE launch (q_actor_step_kernel): critics forward -> objective (the twins
    // exchange q) -> their chain to the action columns -> head backward + the actor's chain
    QActorStep step{};
    int launch_nets = 0;
    step.fwd = critics_forward_args(a.d_critics, cs, used, X3, ldx, B, Bp, c_h1, c_h2, q, nullptr,
                                    nullptr, &launch_nets, critics_img);
    step.fwd.xq = xq + 128;              // the critics' q: lines 4, 5 of the tile
    step.bwd = critics_chain_args(a.d_critics, cs, used, B, Bp, c_h1, c_h2, dq, dh2, dh1, dxa,
                                  &objective, critics_img);
    step.bwd.exchange_failed = failed;
    step.bwd.skip_dz = 1;            // (the critics are frozen in the actor step: only dxa leaves their chain)
    step.bwd.l_q = xq + 128; step.bwd.l_q_at = ValueLines{32, kExchangeTileFloats};
    step.actor = actor_chain_args(a.d_actor, as, B, p_h1 + hs, p_h2 + hs, dloc,
                                  kind == 1 ? dspre : nullptr, ldh, da_h2, da_h1, nullptr, 0, 0, &hb, actor_img);
    step.actor.exchange_failed = failed;
    step.used = used;
    TRY(launch_q_actor_step(step, st));
    TRY(actor_weight_gradients(as, a.d_observations, O, B, p_h1 + hs, p_h2 + hs, dloc,
                               kind == 1 ? dspre : nullptr, ldh, da_h2, da_h1, a.actor.d_grad_sums,
                               st, actor_fold, actor_img));
  } else {
    TRY(critics_forward(a.d_critics, cs, used, X3, ldx, B, Bp, c_h1, c_h2, q, st, nullptr, nullptr, critics_img));
    TRY(critics_backward(a.d_critics, cs, used, X3, ldx, B, Bp, c_h1, c_h2, dq, dh2, dh1, nullptr,
                         dxa, st, &objective, nullptr, critics_img));
    TRY(actor_shaped_backward(a.d_actor, as, a.d_observations, O, B, p_h1 + hs, p_h2 + hs, dloc,
                              kind == 1 ? dspre : nullptr, ldh, da_h2, da_h1, a.actor.d_grad_sums,
                              nullptr, 0, 0, st, &hb, actor_fold, actor_img));
  }
  TONIC_CHECK_LAUNCH("tonic_q_iteration");
  return TONIC_OK;
}

// ------------------------------------------------------------------ D4PG (distributional critic)
// The critic is an actor-shaped network: input [normalised observation | action] (O + A columns),
// two ReLU layers, ONE linear head of NA logits (models/critics.py:49-66) — same packed layout as
// tonic_mlp_actor_param_count(O + A, H, NA, 1).

namespace {

struct DistributionalBuffers {
  float *a_h1, *a_h2, *head, *act, *X, *X2, *t_h1, *t_h2, *t_logits, *c_h1, *c_h2, *logits,
      *dlogits, *loss_m, *dz2, *dz1, *dxa, *dloc, *da_h2, *da_h1;
  static int64_t floats(int Bp, int HP, int ldh, int ldx, int ldl, int A) {
    const int64_t hid = (int64_t)Bp * HP;
    return 2 * hid + (int64_t)Bp * ldh + (int64_t)Bp * A + 2LL * Bp * ldx + 4 * hid +
           3LL * Bp * ldl + Bp + 2 * hid + 2LL * Bp * ldh + 2 * hid + 64 * 24;
  }
  DistributionalBuffers(void* d_workspace, int64_t bytes, int Bp, int HP, int ldh, int ldx,
                        int ldl, int A) {
    Workspace ws{static_cast<char*>(d_workspace), 0, bytes};
    const int64_t hid = (int64_t)Bp * HP;
    a_h1 = ws.take(hid); a_h2 = ws.take(hid);
    head = ws.take((int64_t)Bp * ldh); act = ws.take((int64_t)Bp * A);
    X = ws.take((int64_t)Bp * ldx); X2 = ws.take((int64_t)Bp * ldx);
    t_h1 = ws.take(hid); t_h2 = ws.take(hid); c_h1 = ws.take(hid); c_h2 = ws.take(hid);
    t_logits = ws.take((int64_t)Bp * ldl); logits = ws.take((int64_t)Bp * ldl);
    dlogits = ws.take((int64_t)Bp * ldl); loss_m = ws.take(Bp);
    dz2 = ws.take(hid); dz1 = ws.take(hid);
    dxa = ws.take((int64_t)Bp * ldh); dloc = ws.take((int64_t)Bp * ldh);
    da_h2 = ws.take(hid); da_h1 = ws.take(hid);
  }
};

}  // namespace

extern "C" int64_t tonic_distributional_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H,
                                                        int32_t NA) {
  return DistributionalBuffers::floats(pad16(B), weight_ld(H), pad16(A), pitch16(O + A), pad16(NA),
                                       A) * 4;
}

extern "C" int tonic_distributional_q_grad(
    const float* d_target_actor, const float* d_target_critic, const float* d_critic,
    const float* d_norm_mean, const float* d_norm_std, double norm_clip,
    const float* d_observations, const float* d_actions, const float* d_next_observations,
    const float* d_rewards, const float* d_discounts, const float* d_values, float* d_grad_sums,
    int32_t B, int32_t O, int32_t H, int32_t A, int32_t NA, void* d_workspace,
    int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_target_actor && d_target_critic && d_critic && d_norm_mean && d_norm_std &&
                    d_observations && d_actions && d_next_observations && d_rewards &&
                    d_discounts && d_values && d_grad_sums && d_workspace && B > 0 && NA >= 2 &&
                    NA <= 64,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_distributional_q_grad: bad argument (2 <= atoms <= 64)");
  TONIC_REQUIRE(workspace_bytes >= tonic_distributional_workspace_bytes(B, O, A, H, NA),
                TONIC_ERR_WORKSPACE, "tonic_distributional_q_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), ldl = pad16(NA), threads = 256;
  const DistributionalBuffers w(d_workspace, workspace_bytes, Bp, weight_ld(H), ldh, ldx, ldl, A);
  const ActorShape as{O, H, A, 1}, cs{O + A, H, NA, 1};
  // a' = target_actor(s') (critics.py:104); its tail encodes (s', a') -> X and the stored (s, a) -> X2
  PolicyTail tail{};
  tail.post = POST_COPY; tail.actions = w.act;
  tail.enc_obs = d_next_observations; tail.enc_obs2 = d_observations; tail.enc_act2 = d_actions;
  tail.enc_mean = d_norm_mean; tail.enc_std = d_norm_std; tail.enc_clip = clip_bound(norm_clip);
  tail.enc_out = w.X; tail.enc_out2 = w.X2; tail.enc_ld = ldx;
  bool tail_done = false;
  TRY(actor_forward(d_target_actor, as, d_next_observations, B, w.a_h1, w.a_h2, w.head, w.head, ldh,
                    true, st, &tail, &tail_done));
  if (!tail_done) {
    hipLaunchKernelGGL(copy_actions_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, w.head, ldh, w.act, B, A);
    hipLaunchKernelGGL(encode_kernel, dim3((B * (O + A) + threads - 1) / threads, 2),
                       dim3(threads), 0, st, d_next_observations, w.act, d_norm_mean, d_norm_std,
                       clip_bound(norm_clip), w.X, B, O, A, ldx, d_observations, d_actions, w.X2);
  }
  TRY(actor_forward(d_target_critic, cs, w.X, B, w.t_h1, w.t_h2, w.t_logits, w.t_logits, ldl, false,
                    st, nullptr, nullptr, ldx));
  TRY(actor_forward(d_critic, cs, w.X2, B, w.c_h1, w.c_h2, w.logits, w.logits, ldl, false, st,
                    nullptr, nullptr, ldx));
  hipLaunchKernelGGL(distributional_critic_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, st,
                     w.t_logits, w.logits, ldl, d_rewards, d_discounts, d_values, NA, w.dlogits,
                     w.loss_m, B);
  hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(1024), 0, st, w.loss_m,
                     d_grad_sums + actor_count(cs), B);
  TRY(actor_shaped_backward(d_critic, cs, w.X2, ldx, B, w.c_h1, w.c_h2, w.dlogits, nullptr, ldl,
                            w.dz2, w.dz1, d_grad_sums, nullptr, 0, 0, st));
  TONIC_CHECK_LAUNCH("tonic_distributional_q_grad");
  return TONIC_OK;
}

extern "C" int tonic_distributional_actor_grad(
    const float* d_actor_params, const float* d_critic, const float* d_norm_mean,
    const float* d_norm_std, double norm_clip, const float* d_observations, const float* d_values,
    float* d_grad_sums, int32_t B, int32_t O, int32_t H, int32_t A, int32_t NA, void* d_workspace,
    int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_critic && d_norm_mean && d_norm_std && d_observations &&
                    d_values && d_grad_sums && d_workspace && B > 0 && NA >= 2 && NA <= 64,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_distributional_actor_grad: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_distributional_workspace_bytes(B, O, A, H, NA),
                TONIC_ERR_WORKSPACE, "tonic_distributional_actor_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), ldl = pad16(NA), threads = 256;
  const DistributionalBuffers w(d_workspace, workspace_bytes, Bp, weight_ld(H), ldh, ldx, ldl, A);
  const ActorShape as{O, H, A, 1}, cs{O + A, H, NA, 1};
  PolicyTail tail{};                              // a = actor(s); the tail encodes (s, a) -> X
  tail.post = POST_COPY; tail.actions = w.act;
  tail.enc_obs = d_observations; tail.enc_mean = d_norm_mean; tail.enc_std = d_norm_std;
  tail.enc_clip = clip_bound(norm_clip); tail.enc_out = w.X; tail.enc_ld = ldx;
  bool tail_done = false;
  TRY(actor_forward(d_actor_params, as, d_observations, B, w.a_h1, w.a_h2, w.head, w.head, ldh,
                    true, st, &tail, &tail_done));
  if (!tail_done) {
    hipLaunchKernelGGL(copy_actions_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, w.head, ldh, w.act, B, A);
    hipLaunchKernelGGL(encode_kernel, dim3((B * (O + A) + threads - 1) / threads), dim3(threads),
                       0, st, d_observations, w.act, d_norm_mean, d_norm_std, clip_bound(norm_clip),
                       w.X, B, O, A, ldx);
  }
  TRY(actor_forward(d_critic, cs, w.X, B, w.c_h1, w.c_h2, w.logits, w.logits, ldl, false, st,
                    nullptr, nullptr, ldx));
  hipLaunchKernelGGL(distributional_actor_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, st,
                     w.logits, ldl, d_values, NA, w.dlogits, w.loss_m, B);
  hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(1024), 0, st, w.loss_m,
                     d_grad_sums + actor_count(as), B);
  // through the frozen critic to the action columns of its input, then the tanh head and the actor
  TRY(actor_shaped_backward(d_critic, cs, w.X, ldx, B, w.c_h1, w.c_h2, w.dlogits, nullptr, ldl,
                            w.dz2, w.dz1, nullptr, w.dxa, O, A, st));
  hipLaunchKernelGGL(actor_head_backward_kernel, dim3((B * A + threads - 1) / threads),
                     dim3(threads), 0, st, w.dxa, (const float*)nullptr, ldh, w.act,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ldh, 0.f,
                     0, w.dloc, (float*)nullptr, B, A);
  TRY(actor_shaped_backward(d_actor_params, as, d_observations, O, B, w.a_h1, w.a_h2, w.dloc,
                            nullptr, ldh, w.da_h2, w.da_h1, d_grad_sums, nullptr, 0, 0, st));
  TONIC_CHECK_LAUNCH("tonic_distributional_actor_grad");
  return TONIC_OK;
}


// ------------------------------------------------------------------------------------- MPO
namespace {

struct MpoBuffers {
  float *a_h1, *a_h2, *loc_t, *spre_t, *loc, *spre, *o_h1, *o_h2, *act, *X, *X2, *t_h1, *t_h2, *tq,
      *tq_mean, *c_h1, *c_h2, *q, *dq, *dh2, *dh1, *dloc, *dspre, *da_h2, *da_h1, *part, *klm, *kls;
  static int64_t floats(int B, int S, int HP, int ldh, int ldx, int A) {
    const int64_t Bp = pad16(B), Rp = pad16(S * B);
    return 4 * Bp * HP + 4 * Bp * ldh + Rp * A + Rp * ldx + Bp * ldx + 2 * Rp * HP + Rp + Bp +
           2 * Bp * HP + 2 * Bp + 2 * Bp * HP + 2 * Bp * ldh + 2 * Bp * HP + Bp * kMpoStats +
           2 * Bp * A + 64 * 32;
  }
  MpoBuffers(void* d_workspace, int64_t bytes, int B, int S, int HP, int ldh, int ldx, int A) {
    Workspace ws{static_cast<char*>(d_workspace), 0, bytes};
    const int64_t Bp = pad16(B), Rp = pad16(S * B), hid = Bp * HP;
    a_h1 = ws.take(hid); a_h2 = ws.take(hid); o_h1 = ws.take(hid); o_h2 = ws.take(hid);
    loc_t = ws.take(Bp * ldh); spre_t = ws.take(Bp * ldh); loc = ws.take(Bp * ldh);
    spre = ws.take(Bp * ldh);
    act = ws.take(Rp * A); X = ws.take(Rp * ldx); X2 = ws.take(Bp * ldx);
    t_h1 = ws.take(Rp * HP); t_h2 = ws.take(Rp * HP); tq = ws.take(Rp); tq_mean = ws.take(Bp);
    c_h1 = ws.take(hid); c_h2 = ws.take(hid); q = ws.take(Bp); dq = ws.take(Bp);
    dh2 = ws.take(hid); dh1 = ws.take(hid);
    dloc = ws.take(Bp * ldh); dspre = ws.take(Bp * ldh); da_h2 = ws.take(hid); da_h1 = ws.take(hid);
    part = ws.take(Bp * kMpoStats); klm = ws.take(Bp * A); kls = ws.take(Bp * A);
  }
};

// target_actor(obs) -> S sampled actions per state -> target_critic on the tiled rows: tq [S * B]
int mpo_sampled_values(const float* d_target_actor, const float* d_target_critic,
                       const float* d_norm_mean, const float* d_norm_std, double norm_clip,
                       const float* d_obs, const float* d_eps, int B, int O, int H, int A, int S,
                       const MpoBuffers& w, hipStream_t st) {
  const int ldx = pitch16(O + A), ldh = pad16(A), threads = 256;
  const ActorShape as{O, H, A, 2};
  TRY(actor_forward(d_target_actor, as, d_obs, B, w.a_h1, w.a_h2, w.loc_t, w.spre_t, ldh, true, st));
  const int64_t items = (int64_t)S * B * (O + A);
  hipLaunchKernelGGL(gaussian_tile_kernel, dim3((unsigned)((items + threads - 1) / threads)),
                     dim3(threads), 0, st, d_obs, w.loc_t, w.spre_t, d_eps, ldh, d_norm_mean,
                     d_norm_std, clip_bound(norm_clip), w.act, w.X, B, O, A, S, ldx);
  return critics_forward(d_target_critic, CriticShape{O, A, H}, 1, w.X, ldx, S * B, pad16(S * B),
                         w.t_h1, w.t_h2, w.tq, st);
}

}  // namespace

extern "C" int64_t tonic_mpo_workspace_bytes(int32_t B, int32_t O, int32_t A, int32_t H, int32_t S) {
  return MpoBuffers::floats(B, S, weight_ld(H), pad16(A), pitch16(O + A), A) * 4;
}

extern "C" int tonic_expected_sarsa_grad(
    const float* d_target_actor, const float* d_target_critic, const float* d_critic,
    const float* d_norm_mean, const float* d_norm_std, double norm_clip,
    const float* d_observations, const float* d_actions, const float* d_next_observations,
    const float* d_rewards, const float* d_discounts, const float* d_eps, float* d_grad_sums,
    int32_t B, int32_t O, int32_t H, int32_t A, int32_t S, void* d_workspace,
    int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_target_actor && d_target_critic && d_critic && d_norm_mean && d_norm_std &&
                    d_observations && d_actions && d_next_observations && d_rewards &&
                    d_discounts && d_eps && d_grad_sums && d_workspace && B > 0 && S >= 1 &&
                    S <= kMpoMaxSamples,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_expected_sarsa_grad: bad argument (1 <= samples <= 64)");
  TONIC_REQUIRE(workspace_bytes >= tonic_mpo_workspace_bytes(B, O, A, H, S), TONIC_ERR_WORKSPACE,
                "tonic_expected_sarsa_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), threads = 256;
  const MpoBuffers w(d_workspace, workspace_bytes, B, S, weight_ld(H), pad16(A), ldx, A);
  const CriticShape cs{O, A, H};
  TRY(mpo_sampled_values(d_target_actor, d_target_critic, d_norm_mean, d_norm_std, norm_clip,
                         d_next_observations, d_eps, B, O, H, A, S, w, st));
  hipLaunchKernelGGL(sample_mean_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, st,
                     w.tq, w.tq_mean, B, S);
  hipLaunchKernelGGL(encode_kernel, dim3((B * (O + A) + threads - 1) / threads), dim3(threads), 0,
                     st, d_observations, d_actions, d_norm_mean, d_norm_std, clip_bound(norm_clip),
                     w.X2, B, O, A, ldx);
  TRY(critics_forward(d_critic, cs, 1, w.X2, ldx, B, Bp, w.c_h1, w.c_h2, w.q, st));
  const StepLoss td{LOSS_TD, d_rewards, d_discounts, w.tq_mean, nullptr, 0.f, w.q,
                    d_grad_sums + critic_count(cs)};
  TRY(critics_backward(d_critic, cs, 1, w.X2, ldx, B, Bp, w.c_h1, w.c_h2, w.dq, w.dh2, w.dh1,
                       d_grad_sums, nullptr, st, &td));
  TONIC_CHECK_LAUNCH("tonic_expected_sarsa_grad");
  return TONIC_OK;
}

namespace {
int mpo_actor_grad(
    const float* d_actor_params, const float* d_target_actor, const float* d_target_critic,
    float* d_duals, double min_log_dual, const float* d_norm_mean, const float* d_norm_std, double norm_clip,
    const float* d_observations, const float* d_eps, float* d_grad_sums, float* d_dual_grads,
    float* d_stats, int32_t B, int32_t O, int32_t H, int32_t A, int32_t S, double epsilon,
    double epsilon_penalty, double epsilon_mean, double epsilon_std, int32_t action_penalization,
    void* d_workspace, int64_t workspace_bytes, void* stream, double* d_column_sums) {
  TONIC_REQUIRE(d_actor_params && d_target_actor && d_target_critic && d_duals && d_norm_mean &&
                    d_norm_std && d_observations && d_eps && d_grad_sums &&
                    (d_column_sums || (d_dual_grads && d_stats)) &&
                    d_workspace && B > 0 && S >= 1 && S <= kMpoMaxSamples && A <= 64,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_mpo_actor_grad: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_mpo_workspace_bytes(B, O, A, H, S), TONIC_ERR_WORKSPACE,
                "tonic_mpo_actor_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int ldh = pad16(A);
  const MpoBuffers w(d_workspace, workspace_bytes, B, S, weight_ld(H), ldh, pitch16(O + A), A);
  const ActorShape as{O, H, A, 2};
  TRY(mpo_sampled_values(d_target_actor, d_target_critic, d_norm_mean, d_norm_std, norm_clip,
                         d_observations, d_eps, B, O, H, A, S, w, st));
  TRY(actor_forward(d_actor_params, as, d_observations, B, w.o_h1, w.o_h2, w.loc, w.spre, ldh, true,
                    st));
  hipLaunchKernelGGL(mpo_state_kernel, dim3((B + 3) / 4), dim3(256), 0, st, w.tq,
                     w.act, w.loc_t, w.spre_t, w.loc, w.spre, ldh, d_duals, (float)min_log_dual,
                     action_penalization, w.dloc, w.dspre, w.part, w.klm, w.kls, B, A, S);
  hipLaunchKernelGGL(mpo_dual_kernel, dim3(1), dim3(1024), 0, st, w.part, w.klm, w.kls, d_duals,
                     (float)min_log_dual, action_penalization, (float)epsilon, (float)epsilon_penalty,
                     (float)epsilon_mean, (float)epsilon_std, d_dual_grads, d_stats,
                     d_grad_sums + actor_count(as), B, A, S, (const double*)nullptr, d_column_sums,
                     B);
  TRY(actor_shaped_backward(d_actor_params, as, d_observations, O, B, w.o_h1, w.o_h2, w.dloc,
                            w.dspre, ldh, w.da_h2, w.da_h1, d_grad_sums, nullptr, 0, 0, st));
  TONIC_CHECK_LAUNCH("tonic_mpo_actor_grad");
  return TONIC_OK;
}
}  // namespace

extern "C" int tonic_mpo_actor_grad(
    const float* d_actor_params, const float* d_target_actor, const float* d_target_critic,
    float* d_duals, double min_log_dual, const float* d_norm_mean, const float* d_norm_std, double norm_clip,
    const float* d_observations, const float* d_eps, float* d_grad_sums, float* d_dual_grads,
    float* d_stats, int32_t B, int32_t O, int32_t H, int32_t A, int32_t S, double epsilon,
    double epsilon_penalty, double epsilon_mean, double epsilon_std, int32_t action_penalization,
    void* d_workspace, int64_t workspace_bytes, void* stream) {
  return mpo_actor_grad(d_actor_params, d_target_actor, d_target_critic, d_duals, min_log_dual, d_norm_mean,
                        d_norm_std, norm_clip, d_observations, d_eps, d_grad_sums, d_dual_grads,
                        d_stats, B, O, H, A, S, epsilon, epsilon_penalty, epsilon_mean, epsilon_std,
                        action_penalization, d_workspace, workspace_bytes, stream, nullptr);
}

extern "C" int tonic_mpo_actor_grad_shard(
    const float* d_actor_params, const float* d_target_actor, const float* d_target_critic,
    float* d_duals, double min_log_dual, const float* d_norm_mean, const float* d_norm_std, double norm_clip,
    const float* d_observations, const float* d_eps, float* d_grad_sums, double* d_column_sums,
    int32_t B, int32_t O, int32_t H, int32_t A, int32_t S, int32_t action_penalization,
    void* d_workspace, int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_column_sums != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_mpo_actor_grad_shard: null column sums");
  return mpo_actor_grad(d_actor_params, d_target_actor, d_target_critic, d_duals, min_log_dual, d_norm_mean,
                        d_norm_std, norm_clip, d_observations, d_eps, d_grad_sums, nullptr, nullptr,
                        B, O, H, A, S, 0.0, 0.0, 0.0, 0.0, action_penalization, d_workspace,
                        workspace_bytes, stream, d_column_sums);
}

extern "C" int tonic_mpo_dual_step(const double* d_column_sums, float* d_duals, double min_log_dual,
                                   float* d_dual_grads, float* d_stats, float* d_actor_stats,
                                   int32_t B, int32_t B_global, int32_t A, int32_t S,
                                   double epsilon, double epsilon_penalty, double epsilon_mean,
                                   double epsilon_std, int32_t action_penalization, void* stream) {
  TONIC_REQUIRE(d_column_sums && d_duals && d_dual_grads && d_stats && d_actor_stats && B >= 0 &&
                    B_global > 0 && A >= 1 && A <= 64 && S >= 1,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_mpo_dual_step: bad argument");
  hipLaunchKernelGGL(mpo_dual_kernel, dim3(1), dim3(1024), 0, as_stream(stream),
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, d_duals,
                     (float)min_log_dual, action_penalization, (float)epsilon, (float)epsilon_penalty,
                     (float)epsilon_mean, (float)epsilon_std, d_dual_grads, d_stats, d_actor_stats,
                     B, A, S, d_column_sums, (double*)nullptr, B_global);
  TONIC_CHECK_LAUNCH("tonic_mpo_dual_step");
  return TONIC_OK;
}

// (actors.py:238-267).  Gradient SUMS for the actor + 8 statistics {loss_sum, 0.., B, ..}.
extern "C" int tonic_actor_q_grad(int32_t kind, const float* d_actor_params,
                                  const float* d_critics, const float* d_norm_mean,
                                  const float* d_norm_std, double norm_clip,
                                  const float* d_observations,
                                  const float* d_eps, float* d_grad_sums, int32_t B, int32_t O,
                                  int32_t H, int32_t A, double entropy_coeff, void* d_workspace,
                                  int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_critics && d_norm_mean && d_norm_std && d_observations &&
                    d_grad_sums && d_workspace && B > 0 && (kind == 0 || d_eps),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_actor_q_grad: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_offpolicy_workspace_bytes(B, O, A, H),
                TONIC_ERR_WORKSPACE, "tonic_actor_q_grad: workspace too small");
  hipStream_t st = as_stream(stream);
  const int Bp = pad16(B), ldx = pitch16(O + A), ldh = pad16(A), threads = 256, HP = hidden_pitch(H);
  const int nets = kind == 0 ? 1 : 2;
  const CriticShape cs = critic_shape(O, A, H);
  const ActorShape as = actor_shape(O, H, A, kind == 0 ? 1 : 2);
  const int64_t Pa = actor_count(as);
  Workspace ws{static_cast<char*>(d_workspace), 0, workspace_bytes};
  float* a_h1 = ws.take((int64_t)Bp * HP); float* a_h2 = ws.take((int64_t)Bp * HP);
  float* head0 = ws.take((int64_t)Bp * ldh); float* head1 = ws.take((int64_t)Bp * ldh);
  float* act = ws.take((int64_t)Bp * A); float* sigma = ws.take((int64_t)Bp * A);
  float* logp = ws.take(Bp);
  float* X = ws.take((int64_t)Bp * ldx);
  float* c_h1 = ws.take(2LL * Bp * HP); float* c_h2 = ws.take(2LL * Bp * HP);
  float* q = ws.take(2LL * Bp); float* dq = ws.take(2LL * Bp);
  float* dh2 = ws.take(2LL * Bp * HP); float* dh1 = ws.take(2LL * Bp * HP);
  float* dxa = ws.take(2LL * Bp * ldh);         // action columns of the critics' input gradients
  float* dloc = ws.take((int64_t)Bp * ldh); float* dspre = ws.take((int64_t)Bp * ldh);
  float* da_h2 = ws.take((int64_t)Bp * HP); float* da_h1 = ws.take((int64_t)Bp * HP);
  // the weight images of the actor and the critics (mlpimg.h), formed by this call (see tonic_twin_q_grad)
  ImageSet im = take_images(ws, O, A, H, as.heads, hidden_plain(H) && images_serve(O, A, H) &&
                                                    mlp_forward_supported(H, A, as.heads) &&
                                                    mlp_backward_supported(H, A, as.heads, 0) &&
                                                    mlp_backward_supported(H, 1, 0, A));
  if (im.on) {
    ImgBuild build;
    add_actor_images(build, d_actor_params, as, im.actor);
    add_critic_images(build, d_critics, cs, nets, im.critics.block, im.critics.v);
    TRY(launch_build_images(build, st));
  }

  PolicyTail tail{};                             // the tail also encodes the critics' input (s, a)
  tail.post = kind == 0 ? POST_COPY : POST_SQUASHED_SAMPLE; tail.eps = d_eps; tail.actions = act;
  tail.sigma = kind == 0 ? nullptr : sigma; tail.logp = kind == 0 ? nullptr : logp;
  tail.enc_obs = d_observations; tail.enc_mean = d_norm_mean; tail.enc_std = d_norm_std;
  tail.enc_clip = clip_bound(norm_clip);
  tail.enc_out = X; tail.enc_ld = ldx;
  bool tail_done = false;
  TRY(actor_forward(d_actor_params, as, d_observations, B, a_h1, a_h2, head0, head1, ldh,
                    kind == 0, st, &tail, &tail_done, 0, im.on ? &im.actor : nullptr));
  if (tail_done) {
  } else if (kind == 0) {
    hipLaunchKernelGGL(copy_actions_kernel, dim3((B * A + threads - 1) / threads), dim3(threads),
                       0, st, head0, ldh, act, B, A);
  } else {
    hipLaunchKernelGGL(sac_sample_kernel,
                       dim3((B * sample_group(A) + threads - 1) / threads), dim3(threads), 0, st,
                       head0, head1, d_eps, ldh, act, logp, sigma, B, A,
                       sample_group(A));
  }
  if (!tail_done) {
    hipLaunchKernelGGL(encode_kernel, dim3((B * (O + A) + threads - 1) / threads), dim3(threads),
                       0, st, d_observations, act, d_norm_mean, d_norm_std, clip_bound(norm_clip), X,
                       B, O, A, ldx);
  }
  TRY(critics_forward(d_critics, cs, nets, X, ldx, B, Bp, c_h1, c_h2, q, st, nullptr, nullptr,
                      im.on ? &im.critics : nullptr));
  const StepLoss objective{LOSS_ACTOR, nullptr, nullptr, nullptr, logp, (float)entropy_coeff, q,
                           d_grad_sums + Pa};
  TRY(critics_backward(d_critics, cs, nets, X, ldx, B, Bp, c_h1, c_h2, dq, dh2, dh1, nullptr, dxa,
                       st, &objective, nullptr, im.on ? &im.critics : nullptr));
  hipLaunchKernelGGL(actor_head_backward_kernel, dim3((B * A + threads - 1) / threads),
                     dim3(threads), 0, st, dxa, nets == 2 ? dxa + (int64_t)Bp * ldh : (float*)nullptr,
                     ldh, act, d_eps, sigma, head1, ldh,
                     (float)entropy_coeff, kind == 1 ? 1 : 0, dloc, dspre, B, A);
  TRY(actor_shaped_backward(d_actor_params, as, d_observations, O, B, a_h1, a_h2, dloc,
                            kind == 1 ? dspre : nullptr, ldh, da_h2, da_h1, d_grad_sums, nullptr,
                            0, 0, st, nullptr, nullptr, im.on ? &im.actor : nullptr));
  TONIC_CHECK_LAUNCH("tonic_actor_q_grad");
  return TONIC_OK;
}

extern "C" int tonic_buffer_gather(const int64_t* d_indices, const float* d_buf_observations,
                                   const float* d_buf_actions,
                                   const float* d_buf_next_observations,
                                   const float* d_buf_rewards, const float* d_buf_discounts,
                                   float* d_observations, float* d_actions,
                                   float* d_next_observations, float* d_rewards,
                                   float* d_discounts, int64_t W, int32_t B, int32_t O, int32_t A,
                                   void* stream) {
  TONIC_REQUIRE(d_indices && d_buf_observations && d_buf_actions && d_buf_next_observations &&
                    d_buf_rewards && d_buf_discounts && d_observations && d_actions &&
                    d_next_observations && d_rewards && d_discounts && W > 0 && B > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_buffer_gather: bad argument");
  GatherArgs g{d_indices, d_buf_observations, d_buf_actions, d_buf_next_observations,
               d_buf_rewards, d_buf_discounts, d_observations, d_actions, d_next_observations,
               d_rewards, d_discounts, W, B, O, A};
  hipLaunchKernelGGL(buffer_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), g);
  TONIC_CHECK_LAUNCH("tonic_buffer_gather");
  return TONIC_OK;
}

extern "C" int tonic_buffer_accumulate_n_steps(float* d_buf_next_observations,
                                               float* d_buf_rewards, float* d_buf_discounts,
                                               const float* d_buf_resets,
                                               const float* d_next_observations,
                                               const float* d_rewards,
                                               const float* d_terminations, int64_t row,
                                               int64_t size, int64_t max_size, int64_t W,
                                               int32_t O, int32_t return_steps,
                                               double discount_factor, void* stream) {
  TONIC_REQUIRE(d_buf_next_observations && d_buf_rewards && d_buf_discounts && d_buf_resets &&
                    d_next_observations && d_rewards && d_terminations,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_buffer_accumulate_n_steps: null pointer");
  TONIC_REQUIRE(row >= 0 && row < max_size && size >= 0 && size <= max_size && W > 0 && O > 0 &&
                    return_steps >= 1,
                TONIC_ERR_INVALID_ARGUMENT,
                "tonic_buffer_accumulate_n_steps: row=%lld size=%lld max_size=%lld W=%lld O=%d "
                "return_steps=%d", (long long)row, (long long)size, (long long)max_size,
                (long long)W, O, return_steps);
  const int64_t back = size < return_steps - 1 ? size : return_steps - 1;      // buffers.py:64
  if (back == 0) return TONIC_OK;
  NStepArgs a{d_buf_next_observations, d_buf_rewards, d_buf_discounts, d_buf_resets,
              d_next_observations, d_rewards, d_terminations, row, size, max_size, W, O,
              (int)back, (float)discount_factor};
  const int64_t blocks = (W * O + 255) / 256;
  hipLaunchKernelGGL(buffer_nstep_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
  TONIC_CHECK_LAUNCH("tonic_buffer_accumulate_n_steps");
  return TONIC_OK;
}

extern "C" int tonic_buffer_store(float* d_buf_observations, float* d_buf_actions,
                                  float* d_buf_next_observations, float* d_buf_rewards,
                                  float* d_buf_resets, float* d_buf_terminations,
                                  float* d_buf_discounts, const float* d_observations,
                                  const float* d_actions, const float* d_next_observations,
                                  const float* d_rewards, const float* d_resets,
                                  const float* d_terminations, float* d_norm_acc, int64_t row,
                                  int64_t W, int32_t O, int32_t A, double discount_factor,
                                  void* stream) {
  TONIC_REQUIRE(d_buf_observations && d_buf_actions && d_buf_next_observations &&
                    d_buf_rewards && d_buf_resets && d_buf_terminations && d_buf_discounts &&
                    d_observations && d_actions && d_next_observations && d_rewards && d_resets &&
                    d_terminations && row >= 0 && W > 0 && O > 0 && O <= 1024 && A > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_buffer_store: bad argument");
  BufferStoreArgs a{d_buf_observations, d_buf_actions, d_buf_next_observations, d_buf_rewards,
                    d_buf_resets, d_buf_terminations, d_buf_discounts, d_observations, d_actions,
                    d_next_observations, d_rewards, d_resets, d_terminations, d_norm_acc, row, W,
                    O, A, (float)discount_factor};
  int64_t blocks = (W * (2 * O + A + 4) + 8 * 1024 - 1) / (8 * 1024);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(buffer_store_kernel, dim3((unsigned)blocks), dim3(1024), 0,
                     as_stream(stream), a);
  TONIC_CHECK_LAUNCH("tonic_buffer_store");
  return TONIC_OK;
}
