// Buffer.store row write (tonic/replays/buffers.py:33-52) + MeanStd.record (normalizers/mean_stds.py:44-48) as a
// device function: the body of buffer_store_kernel (offpolicy.hip) and the store role of an acting launch on a
// collector block (mlp_forward_kernel, mlpfwd.hip: the previous step's transition rides in the launch that computes
// the next step's actions — tonic_collector_q_act).
#pragma once
#include "common.h"

namespace tonic {

struct BufferStoreArgs {
  float* b_obs; float* b_act; float* b_next; float* b_rew; float* b_rst; float* b_term; float* b_disc;
  const float* obs; const float* act; const float* next; const float* rew; const float* rst;
  const float* term;
  float* norm_acc;
  int64_t row, W;
  int O, A;
  float discount;
};

#if defined(__HIPCC__)
// `part` of `parts` workgroups; the LAST part also advances the normaliser's running sums (float32, sequential
// row order, square and add as two rounded operations: bit-identical with the reference's Python loop) through an
// LDS tile of `tile_floats` floats.
__device__ __forceinline__ void buffer_store_body(const BufferStoreArgs& a, float* tile, int tile_floats,
                                                  int part, int parts) {
  const int64_t stride = (int64_t)parts * blockDim.x;
  const int64_t tid = (int64_t)part * blockDim.x + threadIdx.x;
  const int64_t n_obs = a.W * a.O, n_act = a.W * a.A;
  for (int64_t i = tid; i < n_obs; i += stride) {
    a.b_obs[a.row * n_obs + i] = a.obs[i];
    a.b_next[a.row * n_obs + i] = a.next[i];
  }
  for (int64_t i = tid; i < n_act; i += stride) a.b_act[a.row * n_act + i] = a.act[i];
  for (int64_t i = tid; i < a.W; i += stride) {
    a.b_rew[a.row * a.W + i] = a.rew[i];
    a.b_rst[a.row * a.W + i] = a.rst[i];
    a.b_term[a.row * a.W + i] = a.term[i];
    a.b_disc[a.row * a.W + i] = (1.f - a.term[i]) * a.discount;     // buffers.py:34-36
  }
  if (a.norm_acc == nullptr || part != parts - 1) return;
  const int64_t rows_per_chunk = tile_floats / a.O;
  for (int k0 = 0; k0 < a.O; k0 += blockDim.x) {                     // (O columns, blockDim.x at a time)
    const int k = k0 + threadIdx.x;
    float sum = 0.f, sum_sq = 0.f;
    if (k < a.O) { sum = a.norm_acc[k]; sum_sq = a.norm_acc[a.O + k]; }
    for (int64_t w0 = 0; w0 < a.W; w0 += rows_per_chunk) {
      const int64_t rows = min(rows_per_chunk, a.W - w0);
      __syncthreads();
      for (int64_t i = threadIdx.x; i < rows * a.O; i += blockDim.x) tile[i] = a.obs[w0 * a.O + i];
      __syncthreads();
      if (k < a.O) record_rows(tile + k, a.O, (int)rows, sum, sum_sq);
    }
    if (k < a.O) { a.norm_acc[k] = sum; a.norm_acc[a.O + k] = sum_sq; }
  }
}
#endif

}  // namespace tonic
