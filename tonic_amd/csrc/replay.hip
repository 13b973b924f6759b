// HBM-resident Segment row store + MeanStd.record (gfx950).
//
// Restates tonic/replays/segments.py:27-36 (Segment.store: one time row of every buffer,
// everything as float32) and tonic/torch/normalizers/mean_stds.py:44-48 (MeanStd.record:
// float32 running sums updated one worker row at a time, `sum_sq += square(val)` as two
// separately rounded operations).  The record is a sequential dependency over workers per
// observation feature, so one lane owns one feature and walks the W rows in order — this is
// bit-exact with the reference by construction (file compiled with -ffp-contract=off).
#include "common.h"

namespace tonic {

struct StoreArgs {
  float* seg_obs; float* seg_act; float* seg_next; float* seg_rew; float* seg_rst;
  float* seg_term; float* seg_lp;
  const float* obs; const float* act; const float* next; const float* rew; const float* rst;
  const float* term; const float* lp;
  float* norm_acc;
  int64_t row, W;
  int O, A;
};

constexpr int kStoreThreads = 1024;
constexpr int kRecordLdsFloats = 16384;     // 64 KiB staging tile for MeanStd.record

__global__ __launch_bounds__(kStoreThreads) void segment_store_kernel(StoreArgs a) {
  __shared__ float tile[kRecordLdsFloats];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_obs = a.W * a.O, n_act = a.W * a.A;
  for (int64_t i = tid; i < n_obs; i += stride) {
    a.seg_obs[a.row * n_obs + i] = a.obs[i];
    a.seg_next[a.row * n_obs + i] = a.next[i];
  }
  for (int64_t i = tid; i < n_act; i += stride) a.seg_act[a.row * n_act + i] = a.act[i];
  for (int64_t i = tid; i < a.W; i += stride) {
    a.seg_rew[a.row * a.W + i] = a.rew[i];
    a.seg_rst[a.row * a.W + i] = a.rst[i];
    a.seg_term[a.row * a.W + i] = a.term[i];
    a.seg_lp[a.row * a.W + i] = a.lp[i];
  }
  // MeanStd.record (mean_stds.py:44-48) on the LAST workgroup: rows are staged through LDS with
  // coalesced loads, then lane k walks feature k over the rows IN ORDER with float32 adds —
  // the reference's exact operation sequence, without a dependent global load per row.
  if (a.norm_acc == nullptr || blockIdx.x != gridDim.x - 1) return;
  const int k = threadIdx.x;
  float sum = 0.f, sum_sq = 0.f;
  if (k < a.O) { sum = a.norm_acc[k]; sum_sq = a.norm_acc[a.O + k]; }
  const int64_t rows_per_chunk = kRecordLdsFloats / a.O;
  for (int64_t w0 = 0; w0 < a.W; w0 += rows_per_chunk) {
    const int64_t rows = min(rows_per_chunk, a.W - w0);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < rows * a.O; i += blockDim.x) tile[i] = a.obs[w0 * a.O + i];
    __syncthreads();
    if (k < a.O) {
      record_rows(tile + k, a.O, (int)rows, sum, sum_sq);
    }
  }
  if (k < a.O) { a.norm_acc[k] = sum; a.norm_acc[a.O + k] = sum_sq; }
}

// MeanStd.record alone (mean_stds.py:44-48) for callers that keep it off the critical path of
// their step kernel (one launch over all the rows of a rollout): one workgroup, {v, v * v}
// staged through LDS, the two chains on two waves IN ROW ORDER — the same float32 sums as the
// reference's Python loop.
__global__ __launch_bounds__(256) void meanstd_record_kernel(const float* values, float* acc,
                                                             int64_t rows, int size) {
  // Two half tiles of {v, v * v}: waves 2..3 stage chunk c+1 while wave 0 walks the sum chain and
  // wave 1 the sum-of-squares chain of chunk c (one dependent add per row and wave).
  constexpr int kQuarter = kRecordLdsFloats / 4;
  __shared__ float tile[2][2][kQuarter];              // [buffer][v | v*v][...]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float chain = 0.f;
  if (wave < 2 && lane < size) chain = acc[wave * size + lane];
  const int64_t per_chunk = kQuarter / size;
  const int64_t chunks = (rows + per_chunk - 1) / per_chunk;
  auto fetch = [&](int64_t c, int first_thread, int threads) {
    const int64_t r0 = c * per_chunk, n = min(per_chunk, rows - r0);
    const float* src = values + r0 * size;
    const int64_t count = n * size;
    // eight loads in flight per thread before the first LDS store: the staging must keep up with
    // a chain that consumes a row every ~5 cycles
    int64_t i = (int)threadIdx.x - first_thread;
    for (; i + 7 * threads < count; i += 8 * threads) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * threads);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        tile[c & 1][0][i + u * threads] = v[u];
        tile[c & 1][1][i + u * threads] = v[u] * v[u];
      }
    }
    for (; i < count; i += threads) {
      const float v = src[i];
      tile[c & 1][0][i] = v;
      tile[c & 1][1][i] = v * v;
    }
  };
  fetch(0, 0, 256);
  __syncthreads();
  for (int64_t c = 0; c < chunks; ++c) {
    if (wave >= 2) {
      if (c + 1 < chunks) fetch(c + 1, 128, 128);
    } else if (lane < size) {
      const int64_t n = min(per_chunk, rows - c * per_chunk);
      add_rows(tile[c & 1][wave] + lane, size, (int)n, chain);
    }
    __syncthreads();
  }
  if (wave < 2 && lane < size) acc[wave * size + lane] = chain;
}

// Segment.get minibatch fetch (segments.py:58-65: `{k: v[indices] for k, v in batch.items()}`)
// for one whole shuffled epoch: wave b copies flattened transition indices[b] of every learner
// input into row b of the contiguous epoch image, so that the minibatches of the epoch are plain
// [start, start + batch_size) slices of it.
struct SegGatherArgs {
  const int64_t* indices;
  const float* obs; const float* act; const float* adv; const float* lp; const float* ret;
  float* o_obs; float* o_act; float* o_adv; float* o_lp; float* o_ret;
  int64_t n, N;
  int O, A;
};

__global__ __launch_bounds__(256) void segment_gather_kernel(SegGatherArgs g) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= g.n) return;
  const int64_t t = g.indices[b];
  if (t < 0 || t >= g.N) return;                       // caller validated; never read outside
  for (int k = lane; k < g.O; k += 64) g.o_obs[b * g.O + k] = g.obs[t * g.O + k];
  for (int k = lane; k < g.A; k += 64) g.o_act[b * g.A + k] = g.act[t * g.A + k];
  if (lane == 0) { g.o_adv[b] = g.adv[t]; g.o_lp[b] = g.lp[t]; g.o_ret[b] = g.ret[t]; }
}

}  // namespace tonic

using namespace tonic;

extern "C" int tonic_meanstd_record(const float* d_values, float* d_norm_acc, int64_t rows,
                                    int32_t size, void* stream) {
  TONIC_REQUIRE(d_values && d_norm_acc && rows > 0 && size > 0 && size <= 64,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_meanstd_record: rows=%lld size=%d",
                (long long)rows, size);
  hipLaunchKernelGGL(meanstd_record_kernel, dim3(1), dim3(256), 0, as_stream(stream), d_values,
                     d_norm_acc, rows, size);
  TONIC_CHECK_LAUNCH("tonic_meanstd_record");
  return TONIC_OK;
}

extern "C" int tonic_segment_gather(const int64_t* d_indices, const float* d_seg_observations,
                                    const float* d_seg_actions, const float* d_seg_advantages,
                                    const float* d_seg_log_probs, const float* d_seg_returns,
                                    float* d_observations, float* d_actions, float* d_advantages,
                                    float* d_log_probs, float* d_returns, int64_t count,
                                    int64_t segment_rows, int32_t O, int32_t A, void* stream) {
  TONIC_REQUIRE(d_indices && d_seg_observations && d_seg_actions && d_seg_advantages &&
                    d_seg_log_probs && d_seg_returns && d_observations && d_actions &&
                    d_advantages && d_log_probs && d_returns,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_segment_gather: null pointer");
  TONIC_REQUIRE(count > 0 && segment_rows > 0 && O > 0 && A > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_segment_gather: count=%lld rows=%lld O=%d A=%d", (long long)count,
                (long long)segment_rows, O, A);
  SegGatherArgs g{d_indices, d_seg_observations, d_seg_actions, d_seg_advantages,
                  d_seg_log_probs, d_seg_returns, d_observations, d_actions, d_advantages,
                  d_log_probs, d_returns, count, segment_rows, O, A};
  hipLaunchKernelGGL(segment_gather_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0,
                     as_stream(stream), g);
  TONIC_CHECK_LAUNCH("tonic_segment_gather");
  return TONIC_OK;
}

extern "C" int tonic_segment_store(float* d_seg_observations, float* d_seg_actions,
                                   float* d_seg_next_observations, float* d_seg_rewards,
                                   float* d_seg_resets, float* d_seg_terminations,
                                   float* d_seg_log_probs, const float* d_observations,
                                   const float* d_actions, const float* d_next_observations,
                                   const float* d_rewards, const float* d_resets,
                                   const float* d_terminations, const float* d_log_probs,
                                   float* d_norm_acc, int64_t row, int64_t W, int32_t O,
                                   int32_t A, void* stream) {
  TONIC_REQUIRE(d_seg_observations && d_seg_actions && d_seg_next_observations &&
                    d_seg_rewards && d_seg_resets && d_seg_terminations && d_seg_log_probs &&
                    d_observations && d_actions && d_next_observations && d_rewards &&
                    d_resets && d_terminations && d_log_probs,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_segment_store: null pointer");
  TONIC_REQUIRE(row >= 0 && W > 0 && O > 0 && O <= 1024 && A > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_segment_store: row=%lld W=%lld O=%d A=%d", (long long)row,
                (long long)W, O, A);
  StoreArgs a;
  a.seg_obs = d_seg_observations; a.seg_act = d_seg_actions; a.seg_next = d_seg_next_observations;
  a.seg_rew = d_seg_rewards; a.seg_rst = d_seg_resets; a.seg_term = d_seg_terminations;
  a.seg_lp = d_seg_log_probs; a.obs = d_observations; a.act = d_actions;
  a.next = d_next_observations; a.rew = d_rewards; a.rst = d_resets; a.term = d_terminations;
  a.lp = d_log_probs; a.norm_acc = d_norm_acc; a.row = row; a.W = W; a.O = O; a.A = A;
  int64_t blocks = (W * (2 * O + A + 4) + 8 * kStoreThreads - 1) / (8 * kStoreThreads);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(segment_store_kernel, dim3((unsigned)blocks), dim3(kStoreThreads), 0,
                     as_stream(stream), a);
  TONIC_CHECK_LAUNCH("tonic_segment_store");
  return TONIC_OK;
}
