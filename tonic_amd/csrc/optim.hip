// Flat fused Adam + PPO statistic finalisation (gfx950).
//
// Restates torch.optim.Adam's single-tensor CPU path (torch/optim/adam.py:395-547; betas
// (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) as constructed by
// tonic/torch/updaters/actors.py:58-59 and critics.py:9-10, applied to ONE flat buffer per
// network (28 B/param of HBM traffic: read p,g,m,v, write p,m,v), plus the tail of
// ClippedRatio.__call__ (actors.py:101-112: loss/kl/entropy/clip_fraction/std/stop) and of
// VRegression.__call__ (critics.py:28).  The optimizer step counter and the PPO early-stop
// flag live on the device so the 80-iteration loop of ppo.py:33-46 needs no host sync.
#include "common.h"

namespace tonic {

struct AdamArgs {
  float* params;
  const float* grad_sums;
  float* exp_avg;
  float* exp_avg_sq;
  int32_t* state;            // {step_count, stop_flag, -, -}
  int64_t n;
  float grad_scale, lr, beta1, beta2, eps;
  double beta1_d, beta2_d, lr_d;
  int stats_kind;            // 0 none, 1 PPO actor, 2 V critic, 3 twin Q critics, 4 Q actor
  float kl_threshold, entropy_coeff;
  const float* adv_stats;
  float* info_row;
  const int32_t* skip;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  if (a.skip != nullptr && *a.skip != 0) return;
  if (a.stats_kind == 1 && a.adv_stats != nullptr && a.adv_stats[2] != 0.f) return;  // actors.py:71
  const int step = a.state[0] + 1;
  const double bias1 = 1.0 - pow(a.beta1_d, (double)step);
  const double bias2 = 1.0 - pow(a.beta2_d, (double)step);
  const float step_size = (float)(a.lr_d / bias1);                 // adam.py:533
  const float bias2_sqrt = (float)sqrt(bias2);                     // adam.py:535
  const float w1 = (float)(1.0 - a.beta1_d), w2 = (float)(1.0 - a.beta2_d);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float g = a.grad_sums[i] * a.grad_scale;
    float m = a.exp_avg[i], v = a.exp_avg_sq[i];
    m = m + w1 * (g - m);                                          // lerp_, adam.py:457
    v = v * a.beta2 + w2 * (g * g);                                // mul_().addcmul_(), :476
    const float denom = sqrtf(v) / bias2_sqrt + a.eps;             // :545
    a.params[i] = a.params[i] - step_size * (m / denom);           // addcdiv_, :547
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
  }
}

// One thread: bump the step counter, turn the statistic sums into the logged values.
__global__ void adam_finalize_kernel(AdamArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;
  const float* st = a.grad_sums + a.n;
  const bool all_zero = a.stats_kind == 1 && a.adv_stats != nullptr && a.adv_stats[2] != 0.f;
  if (!all_zero) a.state[0] += 1;
  if (a.stats_kind == 1 && a.info_row != nullptr) {
    const float entropy = st[3] * a.grad_scale, std = st[4] * a.grad_scale;
    float loss = st[0] * a.grad_scale, kl = st[1] * a.grad_scale;
    float clip_fraction = st[2] * a.grad_scale;
    if (a.entropy_coeff != 0.f) loss -= a.entropy_coeff * entropy;   // actors.py:92-93
    if (all_zero) { loss = 0.f; kl = 0.f; clip_fraction = 0.f; }
    const bool stop = kl > a.kl_threshold;                           // actors.py:112
    a.info_row[0] = loss;
    a.info_row[1] = kl;
    a.info_row[2] = entropy;
    a.info_row[3] = clip_fraction;
    a.info_row[4] = std;
    a.info_row[5] = stop ? 1.f : 0.f;
    a.info_row[6] = 1.f;
    a.info_row[7] = 0.f;
    if (stop) a.state[1] = 1;
  } else if (a.stats_kind == 2 && a.info_row != nullptr) {
    a.info_row[0] = st[0] * a.grad_scale;      // MSE loss
    a.info_row[1] = st[1] * a.grad_scale;      // mean of the pre-step values ('v')
    a.info_row[6] = 1.f;
  } else if (a.stats_kind == 3 && a.info_row != nullptr) {
    a.info_row[0] = st[0] * a.grad_scale;      // loss_1 + loss_2 (critics.py:172,224)
    a.info_row[1] = st[1] * a.grad_scale;      // mean q1
    a.info_row[2] = st[2] * a.grad_scale;      // mean q2
    a.info_row[6] = 1.f;
  } else if (a.stats_kind == 4 && a.info_row != nullptr) {
    a.info_row[0] = st[0] * a.grad_scale;      // actor loss (actors.py:179,257)
    a.info_row[6] = 1.f;
  }
}

// t = t*(1-c) + c*o with three roundings (actor_critics.py:126-130); used by SAC/TD3.
__global__ __launch_bounds__(256) void polyak_kernel(float* target, const float* online,
                                                     int64_t n, float keep, float mix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float scaled = target[i] * keep;
    const float add = mix * online[i];
    target[i] = scaled + add;
  }
}

}  // namespace tonic

using namespace tonic;

extern "C" int tonic_adam_step(float* d_params, const float* d_grad_sums, float* d_exp_avg,
                               float* d_exp_avg_sq, int32_t* d_state, int64_t param_count,
                               double grad_scale, double lr, double beta1, double beta2,
                               double eps, int32_t stats_kind, double kl_threshold,
                               double entropy_coeff,
                               const float* d_adv_stats, float* d_info_row,
                               const int32_t* d_skip_flag, void* stream) {
  TONIC_REQUIRE(d_params && d_grad_sums && d_exp_avg && d_exp_avg_sq && d_state &&
                    param_count > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_adam_step: bad argument");
  TONIC_REQUIRE(stats_kind >= 0 && stats_kind <= 4, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_adam_step: stats_kind %d", stats_kind);
  AdamArgs a;
  a.params = d_params; a.grad_sums = d_grad_sums; a.exp_avg = d_exp_avg;
  a.exp_avg_sq = d_exp_avg_sq; a.state = d_state; a.n = param_count;
  // Hyper-parameters are Python floats in the reference: bias corrections and step size are
  // formed in float64 and only then rounded to float32 (adam.py:530-547).
  a.grad_scale = (float)grad_scale; a.lr = (float)lr; a.beta1 = (float)beta1;
  a.beta2 = (float)beta2; a.eps = (float)eps;
  a.beta1_d = beta1; a.beta2_d = beta2; a.lr_d = lr;
  a.stats_kind = stats_kind; a.kl_threshold = (float)kl_threshold;
  a.entropy_coeff = (float)entropy_coeff;
  a.adv_stats = d_adv_stats; a.info_row = d_info_row; a.skip = d_skip_flag;
  int64_t blocks = (param_count + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(adam_finalize_kernel, dim3(1), dim3(64), 0, st, a);
  TONIC_CHECK_LAUNCH("tonic_adam_step");
  return TONIC_OK;
}

extern "C" int tonic_polyak_update(float* d_target, const float* d_online, int64_t n,
                                   double coeff, void* stream) {
  TONIC_REQUIRE(d_target && d_online && n > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_polyak_update: bad argument");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     d_target, d_online, n, (float)(1.0 - coeff), (float)coeff);
  TONIC_CHECK_LAUNCH("tonic_polyak_update");
  return TONIC_OK;
}
