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

__global__ __launch_bounds__(256) void segment_store_kernel(StoreArgs a) {
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
  // MeanStd.record on the LAST block so it overlaps the copies of the others.
  if (a.norm_acc != nullptr && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < a.O) {
    const int k = threadIdx.x;
    float sum = a.norm_acc[k], sum_sq = a.norm_acc[a.O + k];
    for (int64_t w = 0; w < a.W; ++w) {
      const float v = a.obs[w * a.O + k];
      sum = sum + v;                 // mean_stds.py:46
      const float sq = v * v;        // np.square, then a separate add (:47)
      sum_sq = sum_sq + sq;
    }
    a.norm_acc[k] = sum;
    a.norm_acc[a.O + k] = sum_sq;
  }
}

}  // namespace tonic

using namespace tonic;

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
  TONIC_REQUIRE(row >= 0 && W > 0 && O > 0 && O <= 256 && A > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_segment_store: row=%lld W=%lld O=%d A=%d", (long long)row,
                (long long)W, O, A);
  StoreArgs a;
  a.seg_obs = d_seg_observations; a.seg_act = d_seg_actions; a.seg_next = d_seg_next_observations;
  a.seg_rew = d_seg_rewards; a.seg_rst = d_seg_resets; a.seg_term = d_seg_terminations;
  a.seg_lp = d_seg_log_probs; a.obs = d_observations; a.act = d_actions;
  a.next = d_next_observations; a.rew = d_rewards; a.rst = d_resets; a.term = d_terminations;
  a.lp = d_log_probs; a.norm_acc = d_norm_acc; a.row = row; a.W = W; a.O = O; a.A = A;
  int64_t blocks = (W * O + 255) / 256 + 1;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(segment_store_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     as_stream(stream), a);
  TONIC_CHECK_LAUNCH("tonic_segment_store");
  return TONIC_OK;
}
