// Lambda-return / GAE scan + advantage statistics (HBM-bound), gfx950.
//
// Restates tonic/replays/utils.py:4-19 (reverse scan over T, independent per worker column)
// and tonic/replays/segments.py:41-46 (advantages = returns - values, global mean and
// population std).  Layout: every array is [T, W] row-major float32, so a time row is W
// contiguous floats and lane <-> worker gives perfectly coalesced row accesses.
//
// Parallelisation: one lane per worker column, and the T axis split into `chunks` pieces
// that run concurrently:
//   pass 1 (chunks > 1 only)  each chunk composes its steps into one affine map
//                             ret_in -> A + B * ret_in  (SURVEY.md Appendix A.1);
//   pass 2 (chunks > 1 only)  per column, the carries entering each chunk are formed by
//                             applying the chunk maps from the last chunk backwards;
//   pass 3                    each chunk re-runs the EXACT reference recurrence (same float32
//                             operation order, compiled with -ffp-contract=off) from its
//                             carry, writing returns and raw advantages and accumulating
//                             sum(adv), sum(adv^2) in float64.
// With chunks == 1 only pass 3 runs and the returns are bit-identical to the reference.
#include "common.h"

namespace tonic {

constexpr int kGaeThreads = 256;
constexpr int kUnroll = 8;        // time rows loaded ahead of the dependent chain

struct GaeArgs {
  const float* next_values;
  const float* rewards;
  const float* resets;
  const float* terminations;
  const float* values;
  float* returns;
  float* advantages;
  float* carry_a;     // [chunks, W]
  float* carry_b;     // [chunks, W]
  float* carry_in;    // [chunks, W]
  double* block_sums; // [blocks, 4] = {sum, sum_sq, min, max}
  int64_t T, W;
  int chunks;
  int64_t chunk_len;
  float gamma, lambda, one_minus_lambda;
};

// pass 1: affine summary of one chunk for one column.
__global__ __launch_bounds__(kGaeThreads) void gae_chunk_summary_kernel(GaeArgs g) {
  const int64_t w = (int64_t)blockIdx.x * kGaeThreads + threadIdx.x;
  const int chunk = blockIdx.y;
  if (w >= g.W) return;
  const int64_t t0 = (int64_t)chunk * g.chunk_len;
  const int64_t t1 = min(t0 + g.chunk_len, g.T);
  // ret[t] = A[t] + B[t] * ret[t+1];  compose from the end of the chunk backwards.
  float ca = 0.f, cb = 1.f;
  for (int64_t t = t1 - 1; t >= t0; --t) {
    const int64_t i = t * g.W + w;
    const float nv = g.next_values[i], r = g.rewards[i];
    const float keep = 1.f - g.terminations[i], cont = 1.f - g.resets[i];
    const float b = g.gamma * g.lambda * keep * cont;
    const float a = r + g.gamma * keep * nv * (cont * g.one_minus_lambda + g.resets[i]);
    // (a, b) o (ca, cb) = (a + b * ca, b * cb)
    ca = a + b * ca;
    cb = b * cb;
  }
  g.carry_a[(int64_t)chunk * g.W + w] = ca;
  g.carry_b[(int64_t)chunk * g.W + w] = cb;
}

// pass 2: carry entering each chunk (the value of ret just after the chunk's last step).
__global__ __launch_bounds__(kGaeThreads) void gae_chunk_carry_kernel(GaeArgs g) {
  const int64_t w = (int64_t)blockIdx.x * kGaeThreads + threadIdx.x;
  if (w >= g.W) return;
  float carry = g.next_values[(g.T - 1) * g.W + w];      // utils.py:11
  for (int chunk = g.chunks - 1; chunk >= 0; --chunk) {
    g.carry_in[(int64_t)chunk * g.W + w] = carry;
    carry = g.carry_a[(int64_t)chunk * g.W + w] + g.carry_b[(int64_t)chunk * g.W + w] * carry;
  }
}

// pass 3: the reference recurrence inside one chunk, kUnroll rows of loads in flight.
__global__ __launch_bounds__(kGaeThreads) void gae_scan_kernel(GaeArgs g) {
  const int64_t w = (int64_t)blockIdx.x * kGaeThreads + threadIdx.x;
  const int chunk = blockIdx.y;
  const bool active = w < g.W;
  double sum = 0.0, sum_sq = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  if (active) {
    const int64_t t0 = (int64_t)chunk * g.chunk_len;
    const int64_t t1 = min(t0 + g.chunk_len, g.T);
    float last = g.chunks > 1 ? g.carry_in[(int64_t)chunk * g.W + w]
                              : g.next_values[(g.T - 1) * g.W + w];
    int64_t t = t1 - 1;
    for (; t - (kUnroll - 1) >= t0; t -= kUnroll) {
      float nv[kUnroll], r[kUnroll], rs[kUnroll], tm[kUnroll], v[kUnroll], ret[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (t - u) * g.W + w;
        nv[u] = __builtin_nontemporal_load(g.next_values + i);
        r[u] = __builtin_nontemporal_load(g.rewards + i);
        rs[u] = __builtin_nontemporal_load(g.resets + i);
        tm[u] = __builtin_nontemporal_load(g.terminations + i);
        v[u] = __builtin_nontemporal_load(g.values + i);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        float boot = g.one_minus_lambda * nv[u] + g.lambda * last;   // utils.py:13-14
        boot = boot * (1.f - rs[u]);                                 // :15
        boot = boot + rs[u] * nv[u];                                 // :16
        boot = boot * (1.f - tm[u]);                                 // :17
        last = r[u] + g.gamma * boot;                                // :18
        ret[u] = last;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (t - u) * g.W + w;
        const float adv = ret[u] - v[u];                             // segments.py:42
        __builtin_nontemporal_store(ret[u], g.returns + i);
        __builtin_nontemporal_store(adv, g.advantages + i);
        sum += (double)adv;
        sum_sq += (double)adv * (double)adv;
        lo = fminf(lo, adv);
        hi = fmaxf(hi, adv);
      }
    }
    for (; t >= t0; --t) {
      const int64_t i = t * g.W + w;
      const float nv = g.next_values[i], rs = g.resets[i], tm = g.terminations[i];
      float boot = g.one_minus_lambda * nv + g.lambda * last;
      boot = boot * (1.f - rs);
      boot = boot + rs * nv;
      boot = boot * (1.f - tm);
      last = g.rewards[i] + g.gamma * boot;
      const float adv = last - g.values[i];
      g.returns[i] = last;
      g.advantages[i] = adv;
      sum += (double)adv;
      sum_sq += (double)adv * (double)adv;
      lo = fminf(lo, adv);
      hi = fmaxf(hi, adv);
    }
  }
  // block partial -> HBM (deterministic: fixed lane / wave order, no atomics)
  __shared__ double red[4][kGaeThreads / 64];
  sum = wave_sum(sum);
  sum_sq = wave_sum(sum_sq);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = sum_sq; red[2][wave] = lo; red[3][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < kGaeThreads / 64; ++i) {
      s0 += red[0][i]; s1 += red[1][i];
      mn = fmin(mn, red[2][i]); mx = fmax(mx, red[3][i]);
    }
    const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    g.block_sums[4 * b] = s0;
    g.block_sums[4 * b + 1] = s1;
    g.block_sums[4 * b + 2] = mn;
    g.block_sums[4 * b + 3] = mx;
  }
}

// {mean, std, all_zero, normalise}: segments.py:43-46 and updaters/actors.py:71.
__device__ __forceinline__ void write_adv_stats(double s0, double s1, double mn, double mx,
                                                double count, float* adv_stats) {
  const double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  // numpy's std of a constant array is exactly 0 (segments.py:44 then skips the
  // normalisation); detect that case by min == max instead of trusting var == 0.
  const bool constant = mn == mx;
  const float std = constant ? 0.f : (float)sqrt(var);
  adv_stats[0] = (float)mean;
  adv_stats[1] = std;
  adv_stats[2] = (constant && mn == 0.0) ? 1.f : 0.f;
  adv_stats[3] = std != 0.f ? 1.f : 0.f;
}

// moments = {sum, sum_sq, -min, max, count}: ranks SUM-reduce [0,1,4] and MAX-reduce [2,3].
__global__ void adv_stats_from_moments_kernel(const double* moments, float* adv_stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    write_adv_stats(moments[0], moments[1], -moments[2], moments[3], moments[4], adv_stats);
}

__global__ void gae_stats_kernel(const double* block_sums, int nblocks, double count,
                                 float* adv_stats, double* moments) {
  double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    s0 += block_sums[4 * i];
    s1 += block_sums[4 * i + 1];
    mn = fmin(mn, block_sums[4 * i + 2]);
    mx = fmax(mx, block_sums[4 * i + 3]);
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, off, 64));
    mx = fmax(mx, __shfl_xor(mx, off, 64));
  }
  if (threadIdx.x == 0) {
    write_adv_stats(s0, s1, mn, mx, count, adv_stats);
    if (moments != nullptr) {
      moments[0] = s0; moments[1] = s1; moments[2] = -mn; moments[3] = mx; moments[4] = count;
    }
  }
}

namespace {

int choose_chunks(int64_t T, int64_t W) {
  // Enough (chunk, column) lanes to put several waves on every CU; chunks of >= 32 steps.
  const int64_t want_lanes = 256 * 8 * 64;
  int64_t chunks = 1;
  while (W * chunks < want_lanes && T / (chunks * 2) >= 32) chunks *= 2;
  return (int)chunks;
}

struct GaeLayout {
  int chunks;
  int64_t chunk_len, col_blocks, scan_blocks;
  int64_t off_a, off_b, off_in, off_sums, bytes;
};

GaeLayout gae_layout(int64_t T, int64_t W, int chunks) {
  GaeLayout l;
  l.chunks = chunks > 0 ? chunks : choose_chunks(T, W);
  if (l.chunks > T) l.chunks = (int)(T > 0 ? T : 1);
  l.chunk_len = (T + l.chunks - 1) / l.chunks;
  l.chunks = (int)((T + l.chunk_len - 1) / l.chunk_len);
  l.col_blocks = (W + kGaeThreads - 1) / kGaeThreads;
  l.scan_blocks = l.col_blocks * l.chunks;
  const int64_t cw = round_up((int64_t)l.chunks * W * 4, 256);
  l.off_a = 0;
  l.off_b = cw;
  l.off_in = 2 * cw;
  l.off_sums = 3 * cw;
  l.bytes = l.off_sums + round_up(l.scan_blocks * 32, 256);
  return l;
}

}  // namespace
}  // namespace tonic

using namespace tonic;

extern "C" int64_t tonic_gae_workspace_bytes(int64_t T, int64_t W, int32_t chunks) {
  if (T <= 0 || W <= 0) return 0;
  return gae_layout(T, W, chunks).bytes;
}

extern "C" int tonic_gae_lambda_returns(const float* d_next_values, const float* d_rewards,
                                        const float* d_resets, const float* d_terminations,
                                        const float* d_values, float* d_returns,
                                        float* d_advantages, float* d_adv_stats,
                                        double* d_adv_moments, int64_t T,
                                        int64_t W, double discount_factor, double trace_decay,
                                        int32_t chunks, void* d_workspace,
                                        int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_next_values && d_rewards && d_resets && d_terminations && d_values &&
                    d_returns && d_advantages && d_adv_stats,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_gae_lambda_returns: null pointer");
  TONIC_REQUIRE(T > 0 && W > 0 && chunks >= 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_gae_lambda_returns: T=%lld W=%lld chunks=%d", (long long)T,
                (long long)W, chunks);
  const GaeLayout l = gae_layout(T, W, chunks);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= l.bytes, TONIC_ERR_WORKSPACE,
                "gae workspace too small: %lld < %lld", (long long)workspace_bytes,
                (long long)l.bytes);
  TONIC_REQUIRE(l.chunks <= 65535, TONIC_ERR_INVALID_ARGUMENT, "too many chunks");
  char* ws = static_cast<char*>(d_workspace);
  GaeArgs g;
  g.next_values = d_next_values; g.rewards = d_rewards; g.resets = d_resets;
  g.terminations = d_terminations; g.values = d_values; g.returns = d_returns;
  g.advantages = d_advantages;
  g.carry_a = reinterpret_cast<float*>(ws + l.off_a);
  g.carry_b = reinterpret_cast<float*>(ws + l.off_b);
  g.carry_in = reinterpret_cast<float*>(ws + l.off_in);
  g.block_sums = reinterpret_cast<double*>(ws + l.off_sums);
  g.T = T; g.W = W; g.chunks = l.chunks; g.chunk_len = l.chunk_len;
  // Python-float scalars are rounded to float32 when they multiply a float32 array; `1 -
  // trace_decay` is formed in float64 first (utils.py:14), hence the double arguments.
  g.gamma = (float)discount_factor; g.lambda = (float)trace_decay;
  g.one_minus_lambda = (float)(1.0 - trace_decay);
  hipStream_t st = as_stream(stream);
  if (l.chunks > 1) {
    hipLaunchKernelGGL(gae_chunk_summary_kernel, dim3((unsigned)l.col_blocks, l.chunks),
                       dim3(kGaeThreads), 0, st, g);
    hipLaunchKernelGGL(gae_chunk_carry_kernel, dim3((unsigned)l.col_blocks), dim3(kGaeThreads),
                       0, st, g);
  }
  hipLaunchKernelGGL(gae_scan_kernel, dim3((unsigned)l.col_blocks, l.chunks), dim3(kGaeThreads),
                     0, st, g);
  hipLaunchKernelGGL(gae_stats_kernel, dim3(1), dim3(64), 0, st, g.block_sums,
                     (int)l.scan_blocks, (double)T * (double)W, d_adv_stats, d_adv_moments);
  TONIC_CHECK_LAUNCH("tonic_gae_lambda_returns");
  return TONIC_OK;
}

extern "C" int tonic_advantage_stats_from_moments(const double* d_adv_moments,
                                                  float* d_adv_stats, void* stream) {
  TONIC_REQUIRE(d_adv_moments && d_adv_stats, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_advantage_stats_from_moments: null pointer");
  hipLaunchKernelGGL(adv_stats_from_moments_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                     d_adv_moments, d_adv_stats);
  TONIC_CHECK_LAUNCH("tonic_advantage_stats_from_moments");
  return TONIC_OK;
}
