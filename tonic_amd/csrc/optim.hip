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
  // optional target-network update in the same launch (tonic_adam_polyak_step): `params` is the
  // block [polyak_offset, polyak_offset + n) of the online buffer `polyak_online`
  float* polyak_target;
  const float* polyak_online;
  int64_t polyak_total, polyak_offset;
  float polyak_keep, polyak_mix;
  int adam_blocks;
};

// t = t*(1-c) + c*o with three roundings (actor_critics.py:126-130); used by SAC / TD3 / DDPG.
__device__ __forceinline__ float polyak(float target, float online, float keep, float mix) {
  const float scaled = target * keep;
  const float add = mix * online;
  return scaled + add;
}

// Up to two independent optimizer steps in one launch (blockIdx.y): PPO steps its actor and its
// critic together.
struct AdamPair { AdamArgs net[2]; };

// One thread: bump the step counter, turn the statistic sums into the logged values.
__device__ void adam_finalize(const AdamArgs& a) {
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

__global__ __launch_bounds__(256) void adam_kernel(AdamPair pair) {
  const AdamArgs& a = blockIdx.y == 0 ? pair.net[0] : pair.net[1];
  if (a.polyak_target != nullptr && (int)blockIdx.x >= a.adam_blocks) {
    // the target entries OUTSIDE this optimizer block: their online values are final already
    const int64_t first = (int64_t)((int)blockIdx.x - a.adam_blocks) * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)((int)gridDim.x - a.adam_blocks) * blockDim.x;
    for (int64_t i = first; i < a.polyak_total; i += stride) {
      if (i >= a.polyak_offset && i < a.polyak_offset + a.n) continue;
      a.polyak_target[i] = polyak(a.polyak_target[i], a.polyak_online[i], a.polyak_keep, a.polyak_mix);
    }
    return;
  }
  if (a.skip != nullptr && *a.skip != 0) return;
  const int64_t grid = a.polyak_target != nullptr ? a.adam_blocks : (int64_t)gridDim.x;
  if ((int64_t)blockIdx.x >= grid) return;         // (pairs: the shorter network's spare blocks)
  const bool all_zero = a.stats_kind == 1 && a.adv_stats != nullptr && a.adv_stats[2] != 0.f;  // actors.py:71
  const int step = a.state[0] + 1;
  if (!all_zero) {
    const double bias1 = 1.0 - pow(a.beta1_d, (double)step);
    const double bias2 = 1.0 - pow(a.beta2_d, (double)step);
    const float step_size = (float)(a.lr_d / bias1);                 // adam.py:533
    const float bias2_sqrt = (float)sqrt(bias2);                     // adam.py:535
    const float w1 = (float)(1.0 - a.beta1_d), w2 = (float)(1.0 - a.beta2_d);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += grid * blockDim.x) {
      const float g = a.grad_sums[i] * a.grad_scale;
      float m = a.exp_avg[i], v = a.exp_avg_sq[i];
      m = m + w1 * (g - m);                                          // lerp_, adam.py:457
      v = v * a.beta2 + w2 * (g * g);                                // mul_().addcmul_(), :476
      const float denom = sqrtf(v) / bias2_sqrt + a.eps;             // :545
      const float p = a.params[i] - step_size * (m / denom);         // addcdiv_, :547
      a.params[i] = p;
      a.exp_avg[i] = m;
      a.exp_avg_sq[i] = v;
      if (a.polyak_target != nullptr) {                               // this entry's target, same thread
        float* t = a.polyak_target + a.polyak_offset + i;
        *t = polyak(*t, p, a.polyak_keep, a.polyak_mix);
      }
    }
  }
  // The LAST optimizer workgroup to get here finalises (step counter, logged statistics, KL stop
  // flag) — it used to be a launch of its own (4.6 us for an 8-float row).  Every workgroup has
  // read state[0] / the skip flag before it arrives, so the writes below race with nobody: the
  // barrier in front of the arrival is a workgroup-scope fence (s_waitcnt vmcnt(0) in every wave:
  // all of this workgroup's loads have returned and its stores are acknowledged by L2) and the
  // counter is an agent-scope atomic performed at L2; what the finaliser READS was written by
  // earlier launches.  (A release / acquire pair at agent scope instead would be an L2 write-back
  // + invalidate per workgroup.)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* arrivals = reinterpret_cast<unsigned*>(a.state + 3);
    const unsigned before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    if (before == (unsigned)grid - 1) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      adam_finalize(a);
    }
  }
}

// ---- gradient-norm clipping: torch.nn.utils.clip_grad_norm_ on the flat gradient-sum block ----
// Pass 1: fixed-order partial sums of squares (float64) of a contiguous slice per workgroup.
constexpr int kClipBlocks = 64;

__global__ __launch_bounds__(256) void clip_partials_kernel(const float* sums, int64_t n,
                                                            double* partials,
                                                            const int32_t* skip) {
  __shared__ double red[256];
  if (skip != nullptr && *skip != 0) return;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = min(n, lo + per);
  double acc = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double g = (double)sums[i];
    acc += g * g;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) red[threadIdx.x] += red[threadIdx.x + half];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// Pass 2: every workgroup folds the partials in index order (same bits everywhere) into ||g|| and
// the factor min(1, max_norm / (||g|| + 1e-6)), then scales its slice of the sums in place.
__global__ __launch_bounds__(256) void clip_scale_kernel(float* sums, int64_t n,
                                                         const double* partials, int blocks,
                                                         double grad_scale, float max_norm,
                                                         float* report, const int32_t* skip) {
  __shared__ float coef_shared;
  if (skip != nullptr && *skip != 0) return;
  if (threadIdx.x == 0) {
    double total = 0.0;
    for (int b = 0; b < blocks; ++b) total += partials[b];
    const float norm = (float)(sqrt(total) * grad_scale);        // norm of the MEAN gradient
    const float c = max_norm / (norm + 1e-6f);                    // clip_grad.py: clip_coef ...
    coef_shared = c < 1.0f ? c : 1.0f;                            // ... clamped to 1
    if (blockIdx.x == 0 && report != nullptr) { report[0] = coef_shared; report[1] = norm; }
  }
  __syncthreads();
  const float coef = coef_shared;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    sums[i] = sums[i] * coef;
}

__global__ __launch_bounds__(256) void polyak_kernel(float* target, const float* online,
                                                     int64_t n, float keep, float mix) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    target[i] = polyak(target[i], online[i], keep, mix);
}

}  // namespace tonic

using namespace tonic;

namespace {

int adam_fill(AdamArgs& a, const char* what, float* d_params, const float* d_grad_sums,
              float* d_exp_avg, float* d_exp_avg_sq, int32_t* d_state, int64_t param_count,
              double grad_scale, double lr, double beta1, double beta2, double eps,
              int32_t stats_kind, double kl_threshold, double entropy_coeff,
              const float* d_adv_stats, float* d_info_row, const int32_t* d_skip_flag) {
  TONIC_REQUIRE(d_params && d_grad_sums && d_exp_avg && d_exp_avg_sq && d_state &&
                    param_count > 0,
                TONIC_ERR_INVALID_ARGUMENT, "%s: bad argument", what);
  TONIC_REQUIRE(stats_kind >= 0 && stats_kind <= 4, TONIC_ERR_INVALID_ARGUMENT,
                "%s: stats_kind %d", what, stats_kind);
  a = AdamArgs{};
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
  return TONIC_OK;
}

// Few, fat workgroups: every optimizer workgroup ends with one agent-scope atomic on the same
// counter (the last arriver finalises), and those serialise at ~10 ns each — 760 of them cost more
// than the finalisation launch they replaced (11.8 vs 10.3 us at 200 k parameters).
int adam_blocks_for(int64_t param_count) {
  const int64_t blocks = (param_count + 255) / 256;
  return (int)(blocks > 128 ? 128 : blocks);
}

int adam_launch(float* d_params, const float* d_grad_sums, float* d_exp_avg, float* d_exp_avg_sq,
                int32_t* d_state, int64_t param_count, double grad_scale, double lr, double beta1,
                double beta2, double eps, int32_t stats_kind, double kl_threshold,
                double entropy_coeff, const float* d_adv_stats, float* d_info_row,
                const int32_t* d_skip_flag, float* d_target, const float* d_online,
                int64_t total, int64_t offset, double coeff, void* stream, const char* what) {
  AdamPair pair{};
  AdamArgs& a = pair.net[0];
  if (int rc = adam_fill(a, what, d_params, d_grad_sums, d_exp_avg, d_exp_avg_sq, d_state,
                         param_count, grad_scale, lr, beta1, beta2, eps, stats_kind, kl_threshold,
                         entropy_coeff, d_adv_stats, d_info_row, d_skip_flag))
    return rc;
  const int blocks = adam_blocks_for(param_count);
  int64_t extra = 0;
  if (d_target != nullptr) {
    a.polyak_target = d_target; a.polyak_online = d_online; a.polyak_total = total;
    a.polyak_offset = offset; a.polyak_keep = (float)(1.0 - coeff); a.polyak_mix = (float)coeff;
    a.adam_blocks = blocks;
    extra = (total - param_count + 255) / 256;
    if (extra > 2048) extra = 2048;
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks + extra)), dim3(256), 0, st, pair);
  TONIC_CHECK_LAUNCH(what);
  return TONIC_OK;
}

}  // namespace

extern "C" int tonic_adam_step_pair(
    float* d_params_a, const float* d_grad_sums_a, float* d_exp_avg_a, float* d_exp_avg_sq_a,
    int32_t* d_state_a, int64_t param_count_a, double lr_a, int32_t stats_kind_a,
    double kl_threshold, double entropy_coeff, const float* d_adv_stats, float* d_info_row_a,
    const int32_t* d_skip_flag_a,
    float* d_params_b, const float* d_grad_sums_b, float* d_exp_avg_b, float* d_exp_avg_sq_b,
    int32_t* d_state_b, int64_t param_count_b, double lr_b, int32_t stats_kind_b,
    float* d_info_row_b,
    double grad_scale, double beta1, double beta2, double eps, void* stream) {
  AdamPair pair{};
  if (int rc = adam_fill(pair.net[0], "tonic_adam_step_pair (first)", d_params_a, d_grad_sums_a,
                         d_exp_avg_a, d_exp_avg_sq_a, d_state_a, param_count_a, grad_scale, lr_a,
                         beta1, beta2, eps, stats_kind_a, kl_threshold, entropy_coeff, d_adv_stats,
                         d_info_row_a, d_skip_flag_a))
    return rc;
  if (int rc = adam_fill(pair.net[1], "tonic_adam_step_pair (second)", d_params_b, d_grad_sums_b,
                         d_exp_avg_b, d_exp_avg_sq_b, d_state_b, param_count_b, grad_scale, lr_b,
                         beta1, beta2, eps, stats_kind_b, 0.0, 0.0, nullptr, d_info_row_b, nullptr))
    return rc;
  const int blocks = adam_blocks_for(param_count_a > param_count_b ? param_count_a : param_count_b);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks, 2), dim3(256), 0, st, pair);
  TONIC_CHECK_LAUNCH("tonic_adam_step_pair");
  return TONIC_OK;
}

extern "C" int tonic_adam_step(float* d_params, const float* d_grad_sums, float* d_exp_avg,
                               float* d_exp_avg_sq, int32_t* d_state, int64_t param_count,
                               double grad_scale, double lr, double beta1, double beta2,
                               double eps, int32_t stats_kind, double kl_threshold,
                               double entropy_coeff,
                               const float* d_adv_stats, float* d_info_row,
                               const int32_t* d_skip_flag, void* stream) {
  return adam_launch(d_params, d_grad_sums, d_exp_avg, d_exp_avg_sq, d_state, param_count,
                     grad_scale, lr, beta1, beta2, eps, stats_kind, kl_threshold, entropy_coeff,
                     d_adv_stats, d_info_row, d_skip_flag, nullptr, nullptr, 0, 0, 0.0, stream,
                     "tonic_adam_step");
}

extern "C" int tonic_adam_polyak_step(float* d_online, const float* d_grad_sums, float* d_exp_avg,
                                      float* d_exp_avg_sq, int32_t* d_state, int64_t block_offset,
                                      int64_t param_count, int64_t total_count, double grad_scale,
                                      double lr, double beta1, double beta2, double eps,
                                      int32_t stats_kind, float* d_info_row, float* d_target,
                                      double coeff, void* stream) {
  TONIC_REQUIRE(d_online && d_target && block_offset >= 0 && param_count > 0 &&
                    block_offset + param_count <= total_count,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_adam_polyak_step: block [%lld, +%lld) of %lld",
                (long long)block_offset, (long long)param_count, (long long)total_count);
  return adam_launch(d_online + block_offset, d_grad_sums, d_exp_avg, d_exp_avg_sq, d_state,
                     param_count, grad_scale, lr, beta1, beta2, eps, stats_kind, 0.0, 0.0, nullptr,
                     d_info_row, nullptr, d_target, d_online, total_count, block_offset, coeff,
                     stream, "tonic_adam_polyak_step");
}

extern "C" int64_t tonic_clip_workspace_bytes(int64_t n) {
  (void)n;
  return kClipBlocks * (int64_t)sizeof(double) + 16;
}

extern "C" int tonic_clip_grad_norm(float* d_grad_sums, int64_t n, double grad_scale,
                                    double max_norm, const int32_t* d_skip_flag,
                                    void* d_workspace, int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_grad_sums && d_workspace && n > 0 && max_norm > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_clip_grad_norm: bad argument");
  TONIC_REQUIRE(workspace_bytes >= tonic_clip_workspace_bytes(n), TONIC_ERR_WORKSPACE,
                "tonic_clip_grad_norm: workspace of %lld bytes, %lld needed",
                (long long)workspace_bytes, (long long)tonic_clip_workspace_bytes(n));
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > kClipBlocks) blocks = kClipBlocks;
  double* partials = static_cast<double*>(d_workspace);
  float* report = reinterpret_cast<float*>(partials + kClipBlocks);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(clip_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_grad_sums, n,
                     partials, d_skip_flag);
  int64_t scale_blocks = (n + 255) / 256;
  if (scale_blocks > 1024) scale_blocks = 1024;
  hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)scale_blocks), dim3(256), 0, st,
                     d_grad_sums, n, partials, (int)blocks, grad_scale, (float)max_norm, report,
                     d_skip_flag);
  TONIC_CHECK_LAUNCH("tonic_clip_grad_norm");
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
