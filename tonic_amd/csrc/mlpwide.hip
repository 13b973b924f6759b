// PPO networks OUTSIDE the fused kernels' shapes (observations wider than 32, more than 8 actions):
// the same mathematics, layer by layer.
//
// The fused kernels (mlp64x16.hip) keep a whole 2x64-tanh network per 16-sample tile in registers:
// the layer-1 weight image lives in LDS next to eight per-wave transpose tiles, dW1 sits in 64 * O / 64
// accumulator registers per lane.  Ant-v3 (O = 111), Humanoid (O = 376, A = 17) or humanoid-walk
// (A = 21) fit neither.  Here every layer is its own launch over the whole batch and activations
// travel through HBM (1.5 KB per sample and iteration instead of ~100 B) — about three times the
// fused kernels' time per sample, against the reference's torch-CPU path still two orders of
// magnitude ahead; tonic_ppo_actor_grad / tonic_value_regression_grad / tonic_ppo_act /
// tonic_value_forward dispatch here on their own.
//
//   dense_kernel   Y = act(norm(X) . W^T + b) [* (1 - D^2)]     fp32 MFMA 16x16x4, W staged once per
//                  workgroup into LDS as the MFMA A-operand image (one conflict-free ds_read_b32 per
//                  k-step and feature tile); the transposed form (W^T) serves the backward pass
//                  dZ = (dY . W) * tanh'(H).
//   wgrad_kernel   dW = dY^T . norm(X), db = column sums of dY over a slab of rows per workgroup:
//                  both operands are read straight from HBM in MFMA layout (4 rows x 64 contiguous
//                  bytes per fragment); one partial image per workgroup in the flat parameter layout
//                  of the fused kernels, so reduce_partials_kernel / Adam / clipping are shared.
//   ppo_loss_kernel, value_loss_kernel   the element-wise losses and their gradients with respect
//                  to the head outputs (tonic/torch/updaters/actors.py:81-108, critics.py:18-24).
//   sample_kernel  actions = loc + sigma * eps, log-probabilities (a2c.py:75-85).
#include "mlp64.h"
#include "collect16.h"

namespace tonic {
namespace {

constexpr int kWideThreads = 256;          // 4 waves
constexpr int kWideLd = 32;                // row pitch of the head-sized arrays (A <= 32)
constexpr int kWideBlocks = 1024;          // row slabs = partial images of the weight gradients

__device__ __forceinline__ f32x4 mfma16w(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct DenseArgs {
  const float* X; int ldx;                 // [N, K] input rows
  const float* W; int ldw; int transposed; // weight(j, k) = transposed ? W[k * ldw + j] : W[j * ldw + k]
  const float* bias;                       // [NOUT] or null
  const float* norm_mean; const float* norm_std; float norm_clip;   // null: input not normalised
  const float* D;                          // [N, ldy] or null: multiply the result by act'(z) given D = act(z):
  int dkind;                               //   1 tanh: 1 - D^2;  2 ReLU: D > 0
  float* Y; int ldy;
  int64_t N;
  int K, NOUT, act;                        // act: 0 none, 1 tanh, 2 ReLU
  const int32_t* skip;
};

__device__ __forceinline__ float wide_activation(float z, int act) {
  return act == 1 ? tanh_fast(z) : act == 2 ? fmaxf(z, 0.f) : z;
}
__device__ __forceinline__ float wide_derivative(float v, float d, int dkind) {
  return dkind == 2 ? (d > 0.f ? v : 0.f) : v * (1.f - d * d);
}

// TN = feature tiles of 16 outputs (NOUT <= 16 * TN).
template <int TN>
__global__ __launch_bounds__(kWideThreads) void dense_kernel(DenseArgs a) {
  extern __shared__ float wl[];            // [TN][KS][64] A-operand image of the weights
  if (a.skip != nullptr && *a.skip != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Which input column a lane group contracts at k-step st is free (the weight image follows):
  // plain k = 4 st + g; vec (16-byte aligned rows of a multiple of 16 columns — every hidden-layer
  // input): k = 16 (st / 4) + 4 g + st % 4, so that a lane's four steps are ONE 16-byte load.
  const bool vec = (a.K & 3) == 0 && (a.ldx & 3) == 0;
  const int KS = vec ? 4 * ((a.K + 15) >> 4) : (a.K + 3) >> 2;
  auto column = [&](int st, int gg) { return vec ? 16 * (st >> 2) + 4 * gg + (st & 3) : 4 * st + gg; };
  // Weight image: zero fill, then W read in MEMORY order (coalesced) and scattered to its slot —
  // (j, k) -> tile j / 16, k-step st and lane g * 16 + j % 16 with k = column(st, g).  Sixteen
  // loads in flight per thread: staging is most of a launch over a few hundred rows (the
  // collector's per-step forward), where a gather in image order cost 10 us of 14.
  for (int idx = tid; idx < TN * KS * 64; idx += kWideThreads) wl[idx] = 0.f;
  __syncthreads();
  {
    const int rows = a.transposed ? a.K : a.NOUT, cols = a.transposed ? a.NOUT : a.K;
    const int total = rows * cols;
    for (int base = tid; base < total; base += 16 * kWideThreads) {
      float w[16];
      int slot[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = base + u * kWideThreads;
        const bool live = f < total;
        const int r = live ? f / cols : 0, cc = live ? f - r * cols : 0;
        w[u] = a.W[(int64_t)r * a.ldw + cc];
        const int j = a.transposed ? cc : r, k = a.transposed ? r : cc;
        const int st = vec ? 4 * (k >> 4) + (k & 3) : k >> 2, gg = vec ? (k >> 2) & 3 : k & 3;
        slot[u] = live ? ((j >> 4) * KS + st) * 64 + gg * 16 + (j & 15) : -1;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (slot[u] >= 0) wl[slot[u]] = w[u];
    }
  }
  __syncthreads();
  const int s = lane & 15, g = lane >> 4;
  const int64_t tiles = (a.N + 15) >> 4, stride = (int64_t)gridDim.x * 4;
  // Software pipeline over (tile, chunk of kChunk k-steps): the loads of the NEXT chunk — the
  // next tile's first one at a tile's end — are in flight while the MFMAs of the current chunk
  // run (the plain loop "4 loads, wait, 16 MFMAs" exposed an HBM round trip per 16 input columns).
  // Addresses are clamped and the padding is masked at the consumer, so no load is predicated.
  constexpr int kChunk = 16;
  auto load_chunk = [&](int64_t t, int c0, float (&xv)[kChunk]) {
    const int64_t row = t * 16 + s;
    const float* x = a.X + (row < a.N ? row : a.N - 1) * a.ldx;
    if (vec) {                                         // (uniform) four 16-byte loads
#pragma unroll
      for (int q = 0; q < kChunk / 4; ++q) {
        const int c = (c0 >> 2) + q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (16 * c + 4 * g < a.K ? 16 * c + 4 * g : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[4 * q + e] = v[e];
      }
      return;
    }
    // rows that are not 16-byte aligned (O = 111: 444 bytes): row i of the tile, 64 consecutive
    // columns of the chunk across the lanes — one contiguous 256-byte read per instruction instead
    // of 16 rows x 4 bytes; the consumer transposes through the wave's LDS tile
    const int col = 4 * c0 + lane;
    const int64_t base = t * 16;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int64_t r = base + i < a.N ? base + i : a.N - 1;
      xv[i] = a.X[r * a.ldx + (col < a.K ? col : a.K - 1)];
    }
  };
  constexpr int kTilePitch = 68;                       // (4 s + 4 u + g) mod 64: conflict-free reads
  float* xt = wl + TN * KS * 64 + wave * (16 * kTilePitch);
  float cur[kChunk], nxt[kChunk];
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  int c0 = 0;
  if (t < tiles) load_chunk(t, 0, cur);
  f32x4 acc[TN];
#pragma unroll
  for (int T = 0; T < TN; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
  while (t < tiles) {
    const bool last_chunk = c0 + kChunk >= KS;
    const int64_t t_next = last_chunk ? t + stride : t;
    const int c_next = last_chunk ? 0 : c0 + kChunk;
    if (t_next < tiles) load_chunk(t_next, c_next, nxt);
    const int64_t row = t * 16 + s;
    const bool valid = row < a.N;
    if (!vec) {                                        // (uniform) [row i][column] -> [row s][k-step]
#pragma unroll
      for (int i = 0; i < kChunk; ++i) xt[i * kTilePitch + lane] = cur[i];
      wave_lds_sync();
#pragma unroll
      for (int u = 0; u < kChunk; ++u) cur[u] = xt[s * kTilePitch + 4 * u + g];
      wave_lds_sync();
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int st = c0 + u, k = column(st, g), kc = k < a.K ? k : a.K - 1;
      if (st < KS) {                                   // (uniform)
        float v = cur[u];
        if (a.norm_mean != nullptr)
          v = __builtin_amdgcn_fmed3f((v - a.norm_mean[kc]) / a.norm_std[kc], -a.norm_clip, a.norm_clip);
        v = (valid && k < a.K) ? v : 0.f;
#pragma unroll
        for (int T = 0; T < TN; ++T) acc[T] = mfma16w(wl[(T * KS + st) * 64 + lane], v, acc[T]);
      }
    }
    if (last_chunk) {
      if (valid && (a.NOUT & 3) == 0 && (a.ldy & 3) == 0) {
        // registers 0..3 of a tile are four consecutive outputs of one row: 16-byte stores
#pragma unroll
        for (int T = 0; T < TN; ++T) {
          const int j = 16 * T + 4 * g;
          if (j < a.NOUT) {
            f32x4 v = acc[T];
            if (a.bias != nullptr) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += a.bias[j + r];
            }
            if (a.act != 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = wide_activation(v[r], a.act);
            }
            if (a.D != nullptr) {
              const f32x4 d = *reinterpret_cast<const f32x4*>(a.D + row * a.ldy + j);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = wide_derivative(v[r], d[r], a.dkind);
            }
            *reinterpret_cast<f32x4*>(a.Y + row * a.ldy + j) = v;
          }
        }
      } else if (valid) {
#pragma unroll
        for (int T = 0; T < TN; ++T) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * T + 4 * g + r;              // D[row = feature][col = sample]
            if (j < a.NOUT) {
              float v = acc[T][r] + (a.bias != nullptr ? a.bias[j] : 0.f);
              v = wide_activation(v, a.act);
              if (a.D != nullptr) v = wide_derivative(v, a.D[row * a.ldy + j], a.dkind);
              a.Y[row * a.ldy + j] = v;
            }
          }
        }
      }
#pragma unroll
      for (int T = 0; T < TN; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) cur[u] = nxt[u];
    t = t_next;
    c0 = c_next;
  }
}

struct WgradArgs {
  const float* dY; int ldy; int NOUT;      // [N, ldy]
  const float* X; int ldx; int K;          // [N, ldx]
  const float* norm_mean; const float* norm_std; float norm_clip;
  float* image; int pstride;               // partial images [blocks, pstride]
  int w_offset, b_offset;                  // where W [NOUT, K] and b [NOUT] sit in an image
  int64_t N, slab;                         // rows per workgroup
  const int32_t* skip;
};

// TJ = 16-row tiles of dW (outputs of the layer); every wave owns the column tiles
// tk = wave, wave + 4, ... (at most 6: K <= 384).
template <int TJ>
__global__ __launch_bounds__(kWideThreads) void wgrad_kernel(WgradArgs a) {
  if (a.skip != nullptr && *a.skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g = lane >> 4;
  constexpr int kMaxTk = 6;
  const int TK = (a.K + 15) >> 4;
  f32x4 acc[TJ][kMaxTk];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int i = 0; i < kMaxTk; ++i) acc[tj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) bsum[tj] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * a.slab, r_end = min(a.N, r_begin + a.slab);
  // per-column constants of this wave's tiles (normalisation of the layer-1 input)
  float nm[kMaxTk], ns[kMaxTk];
  int kcol[kMaxTk];
#pragma unroll
  for (int i = 0; i < kMaxTk; ++i) {
    const int k = 16 * (wave + 4 * i) + s;
    kcol[i] = k < a.K ? k : a.K - 1;
    nm[i] = a.norm_mean != nullptr ? a.norm_mean[kcol[i]] : 0.f;
    ns[i] = a.norm_mean != nullptr ? a.norm_std[kcol[i]] : 1.f;
  }
  constexpr int kSteps = 4;                              // k-steps (of 4 samples) loaded together
  // Software pipeline: the loads of the next 16 rows are issued before the MFMAs of the current
  // ones (the plain loop — load, wait, 16 x TJ x tiles MFMAs — exposed an HBM round trip per 16
  // rows of the slab).  Clamped addresses, padding masked at the consumer.
  auto load_rows = [&](int64_t r0, float (&av)[kSteps][TJ], float (&bv)[kSteps][kMaxTk]) {
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
      const int64_t row = r0 + 4 * u + g;                // this lane's sample of the k-step
      const int64_t rc = row < r_end ? row : r_end - 1;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const int j = 16 * tj + s;
        av[u][tj] = a.dY[rc * a.ldy + (j < a.NOUT ? j : 0)];
      }
#pragma unroll
      for (int i = 0; i < kMaxTk; ++i)
        bv[u][i] = wave + 4 * i < TK ? a.X[rc * a.ldx + kcol[i]] : 0.f;
    }
  };
  float av[kSteps][TJ], bv[kSteps][kMaxTk], an[kSteps][TJ], bn[kSteps][kMaxTk];
  if (r_begin < r_end) load_rows(r_begin, av, bv);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 4 * kSteps) {
    if (r0 + 4 * kSteps < r_end) load_rows(r0 + 4 * kSteps, an, bn);
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
      const bool valid = r0 + 4 * u + g < r_end;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
        av[u][tj] = (valid && 16 * tj + s < a.NOUT) ? av[u][tj] : 0.f;
#pragma unroll
      for (int i = 0; i < kMaxTk; ++i) {
        float v = bv[u][i];
        if (a.norm_mean != nullptr)
          v = __builtin_amdgcn_fmed3f((v - nm[i]) / ns[i], -a.norm_clip, a.norm_clip);
        bv[u][i] = (valid && wave + 4 * i < TK && 16 * (wave + 4 * i) + s < a.K) ? v : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kSteps; ++u)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        bsum[tj] += av[u][tj];
#pragma unroll
        for (int i = 0; i < kMaxTk; ++i)
          if (wave + 4 * i < TK) acc[tj][i] = mfma16w(av[u][tj], bv[u][i], acc[tj][i]);
      }
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) av[u][tj] = an[u][tj];
#pragma unroll
      for (int i = 0; i < kMaxTk; ++i) bv[u][i] = bn[u][i];
    }
  }
  float* image = a.image + (int64_t)blockIdx.x * a.pstride;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
#pragma unroll
    for (int i = 0; i < kMaxTk; ++i) {
      const int tk = wave + 4 * i, k = 16 * tk + s;
      if (tk < TK && k < a.K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * tj + 4 * g + r;             // D[row = output j][col = input k]
          if (j < a.NOUT) image[a.w_offset + j * a.K + k] = acc[tj][i][r];
        }
      }
    }
    // bias gradient: this wave saw every sample of the slab; fold the four lane groups
    float b = bsum[tj];
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    if (wave == 0 && g == 0 && 16 * tj + s < a.NOUT) image[a.b_offset + 16 * tj + s] = b;
  }
}

// Row-split form of wgrad_kernel for K <= 16 * TKM columns: every wave takes every fourth round of 16
// rows and ALL TJ x TK tiles (2.5x fewer load instructions per MFMA than the column split, where each
// wave fetched its own copy of dY for one or two column tiles), the four waves' accumulators meet in
// LDS at the end and are added in wave order (bit-reproducible).
template <int TJ, int TKM>
__global__ __launch_bounds__(kWideThreads) void wgrad_rows_kernel(WgradArgs a) {
  extern __shared__ float fold[];                        // [4 waves][TJ * TKM tiles + TJ][64 lanes][4]
  if (a.skip != nullptr && *a.skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, g = lane >> 4;
  const int TK = (a.K + 15) >> 4;
  f32x4 acc[TJ][TKM];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int i = 0; i < TKM; ++i) acc[tj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) bsum[tj] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * a.slab, r_end = min(a.N, r_begin + a.slab);
  float nm[TKM], ns[TKM];
  int kcol[TKM];
#pragma unroll
  for (int i = 0; i < TKM; ++i) {
    const int k = 16 * i + s;
    kcol[i] = k < a.K ? k : a.K - 1;
    nm[i] = a.norm_mean != nullptr ? a.norm_mean[kcol[i]] : 0.f;
    ns[i] = a.norm_mean != nullptr ? a.norm_std[kcol[i]] : 1.f;
  }
  constexpr int kSteps = 4, kRows = 4 * kSteps;
  auto load_rows = [&](int64_t r0, float (&av)[kSteps][TJ], float (&bv)[kSteps][TKM]) {
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
      const int64_t row = r0 + 4 * u + g;
      const int64_t rc = row < r_end ? row : r_end - 1;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const int j = 16 * tj + s;
        av[u][tj] = a.dY[rc * a.ldy + (j < a.NOUT ? j : 0)];
      }
#pragma unroll
      for (int i = 0; i < TKM; ++i) bv[u][i] = i < TK ? a.X[rc * a.ldx + kcol[i]] : 0.f;
    }
  };
  float av[kSteps][TJ], bv[kSteps][TKM], an[kSteps][TJ], bn[kSteps][TKM];
  const int64_t first = r_begin + (int64_t)wave * kRows, step = 4 * kRows;
  if (first < r_end) load_rows(first, av, bv);
  for (int64_t r0 = first; r0 < r_end; r0 += step) {
    if (r0 + step < r_end) load_rows(r0 + step, an, bn);
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
      const bool valid = r0 + 4 * u + g < r_end;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const float v = (valid && 16 * tj + s < a.NOUT) ? av[u][tj] : 0.f;
        bsum[tj] += v;
#pragma unroll
        for (int i = 0; i < TKM; ++i) {
          if (i < TK) {                                  // (uniform)
            float x = bv[u][i];
            if (tj == 0) {
              if (a.norm_mean != nullptr)
                x = __builtin_amdgcn_fmed3f((x - nm[i]) / ns[i], -a.norm_clip, a.norm_clip);
              x = (valid && 16 * i + s < a.K) ? x : 0.f;
              bv[u][i] = x;
            }
            acc[tj][i] = mfma16w(v, x, acc[tj][i]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kSteps; ++u) {
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) av[u][tj] = an[u][tj];
#pragma unroll
      for (int i = 0; i < TKM; ++i) bv[u][i] = bn[u][i];
    }
  }
  // the four waves' accumulators -> LDS, then the sum in wave order
  constexpr int kSlots = TJ * TKM + TJ;
  f32x4* mine = reinterpret_cast<f32x4*>(fold) + (int64_t)wave * kSlots * 64;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
#pragma unroll
    for (int i = 0; i < TKM; ++i) mine[(tj * TKM + i) * 64 + lane] = acc[tj][i];
    float b = bsum[tj];
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    mine[(TJ * TKM + tj) * 64 + lane] = f32x4{b, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  float* image = a.image + (int64_t)blockIdx.x * a.pstride;
  const f32x4* all = reinterpret_cast<const f32x4*>(fold);
  for (int slot = wave; slot < kSlots; slot += 4) {      // each wave folds its share of the slots
    f32x4 v = all[slot * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 o = all[((int64_t)w * kSlots + slot) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += o[r];
    }
    if (slot < TJ * TKM) {
      const int tj = slot / TKM, i = slot - tj * TKM, k = 16 * i + s;
      if (i < TK && k < a.K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * tj + 4 * g + r;             // D[row = output j][col = input k]
          if (j < a.NOUT) image[a.w_offset + j * a.K + k] = v[r];
        }
      }
    } else {
      const int tj = slot - TJ * TKM;
      if (g == 0 && 16 * tj + s < a.NOUT) image[a.b_offset + 16 * tj + s] = v[0];
    }
  }
}

struct PpoLossArgs {
  const float* loc; float* dz3; int ld;    // [N, ld]: tanh'ed head outputs in, d loss / d (pre-tanh) out
  const float* actions; const float* adv; const float* adv_stats; const float* old_logp;
  const float* log_scale;                  // [A] (inside the parameter block)
  float* image; int pstride; int ls_offset; int P;
  int64_t N, slab;
  int A;
  float clip_lo, clip_hi;
  int plain;
  const int32_t* skip;
};

__global__ __launch_bounds__(kWideThreads) void ppo_loss_kernel(PpoLossArgs a) {
  __shared__ float sigma_s[kWideLd], half_inv_var_s[kWideLd], logc_s[kWideLd];
  __shared__ double red[4][kWideLd + 4];
  if (a.skip != nullptr && *a.skip != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < a.A) {
    const float ls = a.log_scale[tid];
    const float sp = ls > 20.f ? ls : log1pf(expf(ls));
    const float sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);       // actors.py:63-64
    sigma_s[tid] = sigma;
    half_inv_var_s[tid] = 1.0f / (2.0f * (sigma * sigma));
    logc_s[tid] = logf(sigma) + kLogSqrt2Pi;
  }
  __syncthreads();
  const float adv_mean = a.adv_stats[0], adv_std = a.adv_stats[1];
  const bool adv_norm = a.adv_stats[3] != 0.f;
  double dsg[kWideLd];
#pragma unroll
  for (int aa = 0; aa < kWideLd; ++aa) dsg[aa] = 0.0;
  double st0 = 0.0, st1 = 0.0, st2 = 0.0, st3 = 0.0;
  const int64_t r_begin = (int64_t)blockIdx.x * a.slab, r_end = min(a.N, r_begin + a.slab);
  for (int64_t n = r_begin + tid; n < r_end; n += kWideThreads) {
    float logp = 0.f;
    for (int aa = 0; aa < a.A; ++aa) {
      const float dif = a.actions[n * a.A + aa] - a.loc[n * a.ld + aa];
      logp += -(dif * dif) * half_inv_var_s[aa] - logc_s[aa];
    }
    const float old_lp = a.old_logp[n];
    float adv = a.adv[n];
    if (adv_norm) adv = (adv - adv_mean) / adv_std;                  // segments.py:45
    const float ratio = __expf(logp - old_lp);
    const float clipped = fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
    const bool outside = ratio > a.clip_hi || ratio < a.clip_lo;
    const bool dead = (ratio > a.clip_hi && adv > 0.f) || (ratio < a.clip_lo && adv < 0.f);
    const bool plain = a.plain != 0;
    const float gl = (plain || !dead) ? -(adv * (plain ? 1.f : ratio)) : 0.f;
    st0 += plain ? -(double)(adv * logp)
                 : -(double)(adv * clipped < adv * ratio ? adv * clipped : adv * ratio);   // torch.min: NaN stays NaN
    st1 += (double)(old_lp - logp);
    st2 += (outside && !plain) ? 1.0 : 0.0;
    st3 += 1.0;
#pragma unroll
    for (int aa = 0; aa < kWideLd; ++aa) {
      if (aa < a.A) {
        const float loc = a.loc[n * a.ld + aa];
        const float dif = a.actions[n * a.A + aa] - loc;
        const float inv_var = 2.f * half_inv_var_s[aa], inv_sigma = 1.f / sigma_s[aa];
        a.dz3[n * a.ld + aa] = gl * dif * inv_var * (1.f - loc * loc);
        dsg[aa] += (double)(gl * (dif * dif * inv_var * inv_sigma - inv_sigma));
      }
    }
  }
  // fixed-order fold: lanes (xor tree), then the four waves in order
  float* image = a.image + (int64_t)blockIdx.x * a.pstride;
#pragma unroll
  for (int aa = 0; aa < kWideLd; ++aa) {
    if (aa < a.A) {
      const double v = wave_sum(dsg[aa]);
      if (lane == 0) red[wave][aa] = v;
    }
  }
  st0 = wave_sum(st0); st1 = wave_sum(st1); st2 = wave_sum(st2); st3 = wave_sum(st3);
  if (lane == 0) {
    red[wave][kWideLd] = st0; red[wave][kWideLd + 1] = st1;
    red[wave][kWideLd + 2] = st2; red[wave][kWideLd + 3] = st3;
  }
  __syncthreads();
  if (tid < a.A)
    image[a.ls_offset + tid] = (float)(((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]);
  if (tid < 8) {
    double v = 0.0;
    const int slot = tid == 0 ? 0 : tid == 1 ? 1 : tid == 2 ? 2 : tid == 5 ? 3 : -1;
    if (slot >= 0)
      v = ((red[0][kWideLd + slot] + red[1][kWideLd + slot]) + red[2][kWideLd + slot]) +
          red[3][kWideLd + slot];
    image[a.P + tid] = (float)v;                        // {loss, kl, clipped, -, -, count, -, -}
  }
}

struct ValueLossArgs {
  const float* values; float* dv; int ld;  // [N, ld], column 0
  const float* returns;
  float* image; int pstride; int P;
  int64_t N, slab;
  const int32_t* skip;
};

__global__ __launch_bounds__(kWideThreads) void value_loss_kernel(ValueLossArgs a) {
  __shared__ double red[4][3];
  if (a.skip != nullptr && *a.skip != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double sq = 0.0, sv = 0.0, cnt = 0.0;
  const int64_t r_begin = (int64_t)blockIdx.x * a.slab, r_end = min(a.N, r_begin + a.slab);
  for (int64_t n = r_begin + tid; n < r_end; n += kWideThreads) {
    const float v = a.values[n * a.ld], err = v - a.returns[n];
    a.dv[n * a.ld] = 2.f * err;                          // d sum((v - ret)^2) / dv (critics.py:22)
    sq += (double)(err * err);
    sv += (double)v;
    cnt += 1.0;
  }
  sq = wave_sum(sq); sv = wave_sum(sv); cnt = wave_sum(cnt);
  if (lane == 0) { red[wave][0] = sq; red[wave][1] = sv; red[wave][2] = cnt; }
  __syncthreads();
  if (tid < 8) {
    float* image = a.image + (int64_t)blockIdx.x * a.pstride;
    const int slot = tid == 0 ? 0 : tid == 1 ? 1 : tid == 5 ? 2 : -1;
    double v = 0.0;
    if (slot >= 0) v = ((red[0][slot] + red[1][slot]) + red[2][slot]) + red[3][slot];
    image[a.P + tid] = (float)v;                        // {sq_err, value, -, -, -, count, -, -}
  }
}

__global__ __launch_bounds__(kWideThreads) void sample_kernel(const float* loc, int ld,
                                                            const float* log_scale,
                                                            const float* eps, float* actions,
                                                            float* log_probs, int64_t n, int A) {
  __shared__ float sigma_s[kWideLd], half_inv_var_s[kWideLd], logc_s[kWideLd];
  if ((int)threadIdx.x < A) {
    const float ls = log_scale[threadIdx.x];
    const float sp = ls > 20.f ? ls : log1pf(expf(ls));
    const float sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);
    sigma_s[threadIdx.x] = sigma;
    half_inv_var_s[threadIdx.x] = 1.0f / (2.0f * (sigma * sigma));
    logc_s[threadIdx.x] = logf(sigma) + kLogSqrt2Pi;
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float logp = 0.f;
    for (int aa = 0; aa < A; ++aa) {
      const float mu = loc[i * ld + aa];
      const float act = eps != nullptr ? mu + sigma_s[aa] * eps[i * A + aa] : mu;   // a2c.py:81
      const float dif = act - mu;
      logp += -(dif * dif) * half_inv_var_s[aa] - logc_s[aa];
      actions[i * A + aa] = act;
    }
    if (log_probs != nullptr) log_probs[i] = logp;
  }
}

__global__ void gather_column_kernel(const float* src, int ld, float* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i * ld];
}

// ---------------------------------------------------------------------------------- host side

int launch_dense_slice(const DenseArgs& a, hipStream_t st);

// One layer: the kernel holds the weight image of at most 64 outputs in LDS — wider layers go out as
// slices of 64 outputs (rows of W, or columns of W^T; bias, mask and result columns move along).
int launch_dense(const DenseArgs& a, hipStream_t st) {
  if (a.NOUT <= 64) return launch_dense_slice(a, st);
  for (int j0 = 0; j0 < a.NOUT; j0 += 64) {
    DenseArgs d = a;
    d.NOUT = a.NOUT - j0 < 64 ? a.NOUT - j0 : 64;
    d.W = a.transposed ? a.W + j0 : a.W + (int64_t)j0 * a.ldw;
    if (a.bias != nullptr) d.bias = a.bias + j0;
    if (a.D != nullptr) d.D = a.D + j0;
    d.Y = a.Y + j0;
    if (int rc = launch_dense_slice(d, st)) return rc;
  }
  return TONIC_OK;
}

int launch_dense_slice(const DenseArgs& a, hipStream_t st) {
  const bool vec = (a.K & 3) == 0 && (a.ldx & 3) == 0;           // (as in the kernel)
  const int tiles_out = (a.NOUT + 15) / 16, ks = vec ? 4 * ((a.K + 15) / 16) : (a.K + 3) / 4;
  // (the kernel is instantiated for 1, 2 or 4 feature tiles and lays its LDS out for THAT many: three tiles
  //  — a 33 .. 48-output slice of a torso layer — run the four-tile instance)
  const int tn = tiles_out <= 1 ? 1 : tiles_out == 2 ? 2 : 4;
  const int lds_bytes = tn * ks * 64 * 4 + (vec ? 0 : 4 * 16 * 68 * 4);   // + the waves' transpose tiles
  const int64_t tiles = (a.N + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  constexpr int kMaxLds = 4 * 96 * 64 * 4 + 4 * 16 * 68 * 4;   // 64 outputs x 384 inputs + transpose tiles
  auto go = [&](auto kernel) {
    static thread_local bool configured = false;           // (one flag per instantiation)
    if (!configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
      if (e != hipSuccess) {
        set_error("mlpwide: %d B of LDS for the weight image: %s", kMaxLds, hipGetErrorString(e));
        return (int)TONIC_ERR_LAUNCH;
      }
      configured = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kWideThreads), lds_bytes, st, a);
    return (int)TONIC_OK;
  };
  int rc;
  if (tn <= 1) rc = go(dense_kernel<1>);
  else if (tn == 2) rc = go(dense_kernel<2>);
  else rc = go(dense_kernel<4>);
  if (rc != TONIC_OK) return rc;
  TONIC_CHECK_LAUNCH("dense_kernel");
  return TONIC_OK;
}

template <int TJ, int TKM>
int launch_wgrad_rows(const WgradArgs& a, int blocks, hipStream_t st) {
  constexpr int lds_bytes = 4 * (TJ * TKM + TJ) * 64 * 16;
  auto kernel = wgrad_rows_kernel<TJ, TKM>;
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("mlpwide: %d B of LDS for the wgrad fold: %s", lds_bytes, hipGetErrorString(e));
      return TONIC_ERR_LAUNCH;
    }
    configured = true;
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kWideThreads), lds_bytes, st, a);
  TONIC_CHECK_LAUNCH("wgrad_rows_kernel");
  return TONIC_OK;
}

int launch_wgrad_slice(const WgradArgs& a, int blocks, hipStream_t st);

// dW of one layer in slices of 64 output rows (the kernels hold at most four 16-row tiles per wave)
int launch_wgrad(const WgradArgs& a, int blocks, hipStream_t st) {
  if (a.NOUT <= 64) return launch_wgrad_slice(a, blocks, st);
  for (int j0 = 0; j0 < a.NOUT; j0 += 64) {
    WgradArgs w = a;
    w.NOUT = a.NOUT - j0 < 64 ? a.NOUT - j0 : 64;
    w.dY = a.dY + j0;
    w.w_offset = a.w_offset + j0 * a.K;
    w.b_offset = a.b_offset + j0;
    if (int rc = launch_wgrad_slice(w, blocks, st)) return rc;
  }
  return TONIC_OK;
}

int launch_wgrad_slice(const WgradArgs& a, int blocks, hipStream_t st) {
  const int tj = (a.NOUT + 15) / 16, tk = (a.K + 15) / 16;
  if (tk <= 4) {                                         // hidden-layer inputs: row split
    if (tj <= 1) return launch_wgrad_rows<1, 4>(a, blocks, st);
    if (tj == 2) return launch_wgrad_rows<2, 4>(a, blocks, st);
    return launch_wgrad_rows<4, 4>(a, blocks, st);
  }
  if (tk <= 7 && tj > 2) return launch_wgrad_rows<4, 7>(a, blocks, st);
  if (tj <= 1) hipLaunchKernelGGL(wgrad_kernel<1>, dim3(blocks), dim3(kWideThreads), 0, st, a);
  else if (tj == 2) hipLaunchKernelGGL(wgrad_kernel<2>, dim3(blocks), dim3(kWideThreads), 0, st, a);
  else hipLaunchKernelGGL(wgrad_kernel<4>, dim3(blocks), dim3(kWideThreads), 0, st, a);
  TONIC_CHECK_LAUNCH("wgrad_kernel");
  return TONIC_OK;
}

struct WideLayout {        // offsets (floats) inside the flat parameter block, and the scratch
  Torso t;
  int W[kMaxTorsoLayers], b[kMaxTorsoLayers], ls, Wh, bh, P;     // parameters() order (models/utils.py:12-23)
  int blocks; int64_t slab, pstride;
  int64_t off_h[kMaxTorsoLayers], off_dz[kMaxTorsoLayers], off_out, off_dzh, off_image, bytes;
  WideLayout(int64_t n, int O, int A, bool actor, const Torso& torso) : t(torso) {
    int at = 0, in = O;
    for (int l = 0; l < t.layers; ++l) {
      W[l] = at; at += t.size[l] * in; b[l] = at; at += t.size[l]; in = t.size[l];
    }
    if (actor) { ls = at; Wh = ls + A; bh = Wh + in * A; P = bh + A; }       // actors.py:52-53: log_scale first
    else { ls = -1; Wh = at; bh = Wh + in; P = bh + 1; }
    blocks = (int)((n + 63) / 64 < kWideBlocks ? (n + 63) / 64 : kWideBlocks);
    if (blocks < 1) blocks = 1;
    slab = round_up((n + blocks - 1) / blocks, 4);
    pstride = round_up(P + kStatSlots, 64);
    int64_t off = 0;
    for (int l = 0; l < t.layers; ++l) { off_h[l] = off; off += round_up(n * t.size[l] * 4, 256); }
    const int64_t head = round_up(n * kWideLd * 4, 256);
    off_out = off; off += head; off_dzh = off; off += head;
    for (int l = 0; l < t.layers; ++l) { off_dz[l] = off; off += round_up(n * t.size[l] * 4, 256); }
    off_image = off;
    bytes = off_image + round_up((int64_t)blocks * pstride * 4, 256);
  }
  int last() const { return t.size[t.layers - 1]; }
};

// forward pass into the scratch: hidden activations, head outputs (tanh'ed locations / the value column)
int wide_forward(const float* params, const WideLayout& L, const float* obs, int64_t n, int O,
                 int A, bool actor, const float* mean, const float* std, float clip, char* ws,
                 const int32_t* skip, hipStream_t st) {
  DenseArgs d{};
  d.N = n; d.skip = skip; d.act = L.t.act;
  d.X = obs; d.ldx = O; d.K = O;
  d.norm_mean = mean; d.norm_std = std; d.norm_clip = clip;
  for (int l = 0; l < L.t.layers; ++l) {
    float* h = reinterpret_cast<float*>(ws + L.off_h[l]);
    d.W = params + L.W[l]; d.ldw = d.K; d.bias = params + L.b[l];
    d.NOUT = L.t.size[l]; d.Y = h; d.ldy = L.t.size[l];
    if (int rc = launch_dense(d, st)) return rc;
    d.norm_mean = nullptr; d.norm_std = nullptr;
    d.X = h; d.ldx = L.t.size[l]; d.K = L.t.size[l];
  }
  d.W = params + L.Wh; d.ldw = d.K; d.bias = params + L.bh; d.NOUT = actor ? A : 1;
  d.Y = reinterpret_cast<float*>(ws + L.off_out); d.ldy = kWideLd;
  d.act = actor ? 1 : 0;                         // loc_activation Tanh (actors.py:44-48) / none (critics.py:11)
  return launch_dense(d, st);
}

// backward pass from the head gradient (in the scratch) to the per-slab partial images
int wide_backward(const float* params, const WideLayout& L, const float* obs, int64_t n, int O,
                  int A, bool actor, const float* mean, const float* std, float clip, char* ws,
                  const int32_t* skip, hipStream_t st) {
  float* image = reinterpret_cast<float*>(ws + L.off_image);
  const int nout = actor ? A : 1;
  WgradArgs w{};
  w.image = image; w.pstride = (int)L.pstride; w.N = n; w.slab = L.slab; w.skip = skip;
  DenseArgs d{};
  d.N = n; d.skip = skip; d.act = 0; d.transposed = 1; d.dkind = L.t.act;
  // the layer above layer l: its gradient dY [n, ldy] of `outs` columns, its weights at `w_up`
  const float* dY = reinterpret_cast<const float*>(ws + L.off_dzh);
  int ldy = kWideLd, outs = nout, w_up = L.Wh, b_up = L.bh;
  for (int l = L.t.layers - 1; l >= 0; --l) {
    const int H = L.t.size[l];
    const float* h = reinterpret_cast<const float*>(ws + L.off_h[l]);
    float* dz = reinterpret_cast<float*>(ws + L.off_dz[l]);
    // weight gradient of the layer above: dW = dY^T . h
    w.dY = dY; w.ldy = ldy; w.NOUT = outs; w.X = h; w.ldx = H; w.K = H;
    w.w_offset = w_up; w.b_offset = b_up;
    if (int rc = launch_wgrad(w, L.blocks, st)) return rc;
    // dz_l = (dY . W_up) * act'(h_l)
    d.X = dY; d.ldx = ldy; d.K = outs; d.W = params + w_up; d.ldw = H; d.NOUT = H;
    d.D = h; d.Y = dz; d.ldy = H;
    if (int rc = launch_dense(d, st)) return rc;
    dY = dz; ldy = H; outs = H; w_up = L.W[l]; b_up = L.b[l];
  }
  w.dY = dY; w.ldy = ldy; w.NOUT = outs; w.X = obs; w.ldx = O; w.K = O;
  w.w_offset = L.W[0]; w.b_offset = L.b[0];
  w.norm_mean = mean; w.norm_std = std; w.norm_clip = clip;
  return launch_wgrad(w, L.blocks, st);
}

}  // namespace

bool wide_shape(int O, int A, bool actor) { return O > 32 || (actor && A > 8); }

bool wide_supported(int O, int A, bool actor) {
  return O >= 1 && O <= 384 && (!actor || (A >= 1 && A <= kWideLd));
}

bool torso_supported(const Torso& t) {
  if (t.layers < 1 || t.layers > kMaxTorsoLayers || (t.act != 1 && t.act != 2)) return false;
  for (int l = 0; l < t.layers; ++l)
    if (t.size[l] < 4 || t.size[l] > 384 || t.size[l] % 4 != 0) return false;
  return true;
}

int64_t torso_param_count(int O, int A, bool actor, const Torso& t) {
  return WideLayout(1, O, A, actor, t).P;
}

int64_t wide_workspace_bytes(int64_t n, int O, int A, bool actor, const Torso& t) {
  return WideLayout(n, O, A, actor, t).bytes;
}

int wide_actor_grad(const MlpArgs& a, float* d_grad_sums, float entropy_coeff, void* d_workspace,
                    int64_t workspace_bytes, hipStream_t st, const Torso& t) {
  const WideLayout L(a.n, a.O, a.A, true, t);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= L.bytes, TONIC_ERR_WORKSPACE,
                "wide actor grad: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
                (long long)L.bytes);
  char* ws = static_cast<char*>(d_workspace);
  if (int rc = wide_forward(a.params, L, a.obs, a.n, a.O, a.A, true, nullptr, nullptr, 0.f, ws,
                            a.skip, st))
    return rc;
  PpoLossArgs l{};
  l.loc = reinterpret_cast<float*>(ws + L.off_out);
  l.dz3 = reinterpret_cast<float*>(ws + L.off_dzh);
  l.ld = kWideLd; l.actions = a.actions; l.adv = a.adv; l.adv_stats = a.adv_stats;
  l.old_logp = a.old_logp; l.log_scale = a.params + L.ls;
  l.image = reinterpret_cast<float*>(ws + L.off_image); l.pstride = (int)L.pstride;
  l.ls_offset = L.ls; l.P = L.P; l.N = a.n; l.slab = L.slab; l.A = a.A;
  l.clip_lo = a.clip_lo; l.clip_hi = a.clip_hi; l.plain = a.plain; l.skip = a.skip;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(L.blocks), dim3(kWideThreads), 0, st, l);
  TONIC_CHECK_LAUNCH("ppo_loss_kernel");
  if (int rc = wide_backward(a.params, L, a.obs, a.n, a.O, a.A, true, nullptr, nullptr, 0.f, ws,
                             a.skip, st))
    return rc;
  return launch_reduce_partials(true, l.image, L.blocks, (int)L.pstride, L.P, a.params,
                                d_grad_sums, a.O, a.A, entropy_coeff, (double)a.n, a.skip, st, L.ls);
}

int wide_critic_grad(const MlpArgs& a, float* d_grad_sums, void* d_workspace,
                     int64_t workspace_bytes, hipStream_t st, const Torso& t) {
  const WideLayout L(a.n, a.O, 1, false, t);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= L.bytes, TONIC_ERR_WORKSPACE,
                "wide critic grad: workspace of %lld bytes, %lld needed",
                (long long)workspace_bytes, (long long)L.bytes);
  char* ws = static_cast<char*>(d_workspace);
  if (int rc = wide_forward(a.params, L, a.obs, a.n, a.O, 1, false, a.norm_mean, a.norm_std,
                            a.norm_clip, ws, a.skip, st))
    return rc;
  ValueLossArgs l{};
  l.values = reinterpret_cast<float*>(ws + L.off_out);
  l.dv = reinterpret_cast<float*>(ws + L.off_dzh);
  l.ld = kWideLd; l.returns = a.returns;
  l.image = reinterpret_cast<float*>(ws + L.off_image); l.pstride = (int)L.pstride; l.P = L.P;
  l.N = a.n; l.slab = L.slab; l.skip = a.skip;
  hipLaunchKernelGGL(value_loss_kernel, dim3(L.blocks), dim3(kWideThreads), 0, st, l);
  TONIC_CHECK_LAUNCH("value_loss_kernel");
  if (int rc = wide_backward(a.params, L, a.obs, a.n, a.O, 1, false, a.norm_mean, a.norm_std,
                             a.norm_clip, ws, a.skip, st))
    return rc;
  return launch_reduce_partials(false, l.image, L.blocks, (int)L.pstride, L.P, a.params,
                                d_grad_sums, a.O, 1, 0.f, (double)a.n, a.skip, st);
}

int wide_act(const MlpArgs& a, void* d_workspace, int64_t workspace_bytes, hipStream_t st,
             const Torso& t) {
  const WideLayout L(a.n, a.O, a.A, true, t);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= L.bytes, TONIC_ERR_WORKSPACE,
                "wide act: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
                (long long)L.bytes);
  char* ws = static_cast<char*>(d_workspace);
  if (int rc = wide_forward(a.params, L, a.obs, a.n, a.O, a.A, true, nullptr, nullptr, 0.f, ws,
                            nullptr, st))
    return rc;
  int64_t blocks = (a.n + kWideThreads - 1) / kWideThreads;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(kWideThreads), 0, st,
                     reinterpret_cast<const float*>(ws + L.off_out), kWideLd, a.params + L.ls,
                     a.eps, a.out0, a.out1, a.n, a.A);
  TONIC_CHECK_LAUNCH("sample_kernel");
  return TONIC_OK;
}

// ------------------------------------------------------------ the collector's step (wide shapes)
namespace {

constexpr int kIngestThreads = 512;
constexpr int kIngestCopyBlocks = 4;

// src (pinned host memory, 16-byte aligned) -> up to two destinations; every thread's loads are in
// flight before its first store (one PCIe round trip per ~32 KB per workgroup)
__device__ __forceinline__ void ingest_copy(const float* src, float* dst0, float* dst1, int64_t count,
                                            int part, int parts) {
  const int64_t vecs = count >> 2, stride = (int64_t)parts * kIngestThreads;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  for (int64_t base = (int64_t)part * kIngestThreads + threadIdx.x; base < vecs; base += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + u * stride < vecs) v[u] = __builtin_nontemporal_load(s4 + base + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = base + u * stride;
      if (i < vecs) {
        if (dst0 != nullptr) reinterpret_cast<f32x4*>(dst0)[i] = v[u];
        if (dst1 != nullptr) reinterpret_cast<f32x4*>(dst1)[i] = v[u];
      }
    }
  }
  if (part == 0 && threadIdx.x < (count & 3)) {           // tail floats
    const int64_t i = (vecs << 2) + threadIdx.x;
    const float v = src[i];
    if (dst0 != nullptr) dst0[i] = v;
    if (dst1 != nullptr) dst1[i] = v;
  }
}

struct IngestArgs {
  WideCollect c;
  float* d_obs; float* d_eps;
};

__global__ __launch_bounds__(kIngestThreads) void wide_ingest_kernel(IngestArgs a) {
  __shared__ float tile[16384];
  const WideCollect& c = a.c;
  const int64_t W = c.W, n_obs = W * c.O, n_eps = W * c.A;
  if ((int)blockIdx.x < kIngestCopyBlocks) {
    const int part = blockIdx.x;
    // this step's inputs: device staging for the forward pass, observations also into their row
    ingest_copy(c.obs, a.d_obs, c.seg_obs + c.row * n_obs, n_obs, part, kIngestCopyBlocks);
    if (c.eps != nullptr) ingest_copy(c.eps, a.d_eps, nullptr, n_eps, part, kIngestCopyBlocks);
    if (c.outcome_row >= 0) {                            // segments.py:27-36, the previous step's
      ingest_copy(c.next_obs, c.seg_next + c.outcome_row * n_obs, nullptr, n_obs, part,
                  kIngestCopyBlocks);
      ingest_copy(c.rewards, c.seg_rew + c.outcome_row * W, nullptr, W, part, kIngestCopyBlocks);
      ingest_copy(c.resets, c.seg_rst + c.outcome_row * W, nullptr, W, part, kIngestCopyBlocks);
      ingest_copy(c.terminations, c.seg_term + c.outcome_row * W, nullptr, W, part,
                  kIngestCopyBlocks);
    }
    return;
  }
  // MeanStd.record (mean_stds.py:44-48): thread k walks feature k over the rows IN ORDER; the sums
  // go from history entry `row` to `row + 1` (collector.hip: a step may be issued twice)
  if (c.norm_hist == nullptr) return;
  const int k = threadIdx.x;
  const float* in = c.norm_hist + c.row * 2 * c.O;
  float* out = c.norm_hist + (c.row + 1) * 2 * c.O;
  float sum = 0.f, sum_sq = 0.f;
  if (k < c.O) { sum = in[k]; sum_sq = in[c.O + k]; }
  const int64_t rows_per_chunk = (16384 / c.O) & ~(int64_t)3;
  for (int64_t w0 = 0; w0 < W; w0 += rows_per_chunk) {
    const int64_t rows = min(rows_per_chunk, W - w0);
    __syncthreads();
    ingest_copy(c.obs + w0 * c.O, tile, nullptr, rows * c.O, 0, 1);
    __syncthreads();
    if (k < c.O) record_rows(tile + k, c.O, (int)rows, sum, sum_sq);
  }
  if (k < c.O) { out[k] = sum; out[c.O + k] = sum_sq; }
}

// a2c.py:75-85 after the head: actions = loc + sigma * eps and their log-probabilities -> the
// Segment row and the block's action field, then the completion word of this workgroup (the
// ingest launch ahead of it on the stream has read everything it needs from the block by now)
__global__ __launch_bounds__(kWideThreads) void wide_sample_store_kernel(
    const float* loc, int ld, const float* log_scale, const float* eps, WideCollect c) {
  __shared__ float sigma_s[kWideLd], half_inv_var_s[kWideLd], logc_s[kWideLd];
  if ((int)threadIdx.x < c.A) {
    const float ls = log_scale[threadIdx.x];
    const float sp = ls > 20.f ? ls : log1pf(expf(ls));
    const float sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);
    sigma_s[threadIdx.x] = sigma;
    half_inv_var_s[threadIdx.x] = 1.0f / (2.0f * (sigma * sigma));
    logc_s[threadIdx.x] = logf(sigma) + kLogSqrt2Pi;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.W) {
    float logp = 0.f;
    for (int aa = 0; aa < c.A; ++aa) {
      const float mu = loc[i * ld + aa];
      const float act = eps != nullptr ? mu + sigma_s[aa] * eps[i * c.A + aa] : mu;   // a2c.py:81
      const float dif = act - mu;
      logp += -(dif * dif) * half_inv_var_s[aa] - logc_s[aa];
      c.seg_act[(c.row * c.W + i) * c.A + aa] = act;
      c.actions_out[i * c.A + aa] = act;
    }
    c.seg_lp[c.row * c.W + i] = logp;
  }
  // pinned host memory is cached write-back in this XCD's L2: release before the word goes out
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (threadIdx.x == 0 && c.done_flags != nullptr)
    __hip_atomic_store(c.done_flags + blockIdx.x, c.done_seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

int wide_collect_words(int64_t W) { return (int)((W + kWideThreads - 1) / kWideThreads); }

int64_t wide_collect_workspace_bytes(int64_t W, int O, int A) {
  return WideLayout(W, O, A, true, Torso::standard()).bytes + round_up(W * O * 4, 256) +
         round_up(W * A * 4, 256);
}

int wide_collect_step(const WideCollect& c, void* d_workspace, int64_t workspace_bytes,
                      hipStream_t st) {
  const WideLayout L(c.W, c.O, c.A, true, Torso::standard());
  TONIC_REQUIRE(d_workspace && workspace_bytes >= wide_collect_workspace_bytes(c.W, c.O, c.A),
                TONIC_ERR_WORKSPACE, "wide collect step: workspace too small");
  char* ws = static_cast<char*>(d_workspace);
  IngestArgs in{c, reinterpret_cast<float*>(ws + L.bytes),
                reinterpret_cast<float*>(ws + L.bytes + round_up(c.W * c.O * 4, 256))};
  hipLaunchKernelGGL(wide_ingest_kernel, dim3(kIngestCopyBlocks + 1), dim3(kIngestThreads), 0, st, in);
  TONIC_CHECK_LAUNCH("wide_ingest_kernel");
  if (int rc = wide_forward(c.params, L, in.d_obs, c.W, c.O, c.A, true, nullptr, nullptr, 0.f, ws,
                            nullptr, st))
    return rc;
  hipLaunchKernelGGL(wide_sample_store_kernel, dim3(wide_collect_words(c.W)), dim3(kWideThreads), 0,
                     st, reinterpret_cast<const float*>(ws + L.off_out), kWideLd, c.params + L.ls,
                     c.eps != nullptr ? in.d_eps : (const float*)nullptr, c);
  TONIC_CHECK_LAUNCH("wide_sample_store_kernel");
  return TONIC_OK;
}

int wide_value(const MlpArgs& a, void* d_workspace, int64_t workspace_bytes, hipStream_t st,
               const Torso& t) {
  const WideLayout L(a.n, a.O, 1, false, t);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= L.bytes, TONIC_ERR_WORKSPACE,
                "wide value forward: workspace of %lld bytes, %lld needed",
                (long long)workspace_bytes, (long long)L.bytes);
  char* ws = static_cast<char*>(d_workspace);
  if (int rc = wide_forward(a.params, L, a.obs, a.n, a.O, 1, false, a.norm_mean, a.norm_std,
                            a.norm_clip, ws, nullptr, st))
    return rc;
  int64_t blocks = (a.n + kWideThreads - 1) / kWideThreads;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gather_column_kernel, dim3((unsigned)blocks), dim3(kWideThreads), 0, st,
                     reinterpret_cast<const float*>(ws + L.off_out), kWideLd, a.out0, a.n);
  TONIC_CHECK_LAUNCH("gather_column_kernel");
  return TONIC_OK;
}

}  // namespace tonic
