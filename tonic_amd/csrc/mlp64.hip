// Fused 2x64-tanh MLP kernels for the PPO actor / V critic (gfx950, wave64, fp32 MFMA).
//
// Math restated from (paths relative to the reference checkout):
//   tonic/torch/models/utils.py:12-23 (MLP), actors.py:60-66,134-137 (Gaussian head),
//   critics.py:15-20,87-90 (value head), encoders.py:13-16, normalizers/mean_stds.py:34-39,
//   tonic/torch/updaters/actors.py:70-112 (ClippedRatio), critics.py:18-28 (VRegression).
//
// Design (one wave = one 32-sample tile, no block-level barriers inside the tile loop):
//   * "S layout": lane = (sample s = lane&31, half h = lane>>5); register q in [0,32) holds
//     hidden feature feat(q,h).  This is exactly the C/D layout of
//     v_mfma_f32_32x32x2_f32 when the product is formed TRANSPOSED,
//     D[out_feature][sample] = sum_k W[out_feature][k] * X^T[k][sample]
//     (A operand = weights streamed from LDS, B operand = activations in registers), and it
//     is ALSO a valid B-operand layout for the next layer because the k order of an MFMA
//     chain is free: step s contracts feature feat(s,0) from lanes 0-31 with feat(s,1) from
//     lanes 32-63, and the LDS weight image is pre-permuted to match.  So the forward chain
//     x -> h1 -> h2 and the backward chain dz2 -> dh1 never leave registers.
//   * Weight gradients contract over SAMPLES, which needs "F layout" operands (lane = feature,
//     registers = samples).  Each wave transposes S -> F through a private LDS scratch
//     (row stride 36 floats: conflict-free ds_write_b32 / ds_read_b128), then accumulates
//     dW2 (4 tiles), dW1 (2 tiles) with MFMA and the skinny dW3 / biases with VALU, all in
//     registers across the wave's whole tile loop.
//   * End of kernel: waves fold their accumulators into one LDS image in wave order
//     (deterministic), one partial block per workgroup goes to HBM, a second tiny kernel
//     reduces the partials in fixed order.
#include <string.h>

#include <type_traits>

#include <atomic>
#include <unordered_set>

#include "mlp64.h"
#include "mlpfwd.h"

namespace tonic { extern std::atomic<int> g_gae_stream; }      // gae.hip (tuning key "gae_stream")

namespace tonic {

constexpr int TS = 36;  // floats per row of the per-wave transpose scratch

__host__ __device__ constexpr int feat(int q, int h) {
  return 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h;
}

template <int KS1, int AP, bool BWD, int WAVES>
struct Lds {
  static constexpr int W1S = 0;                                  // [2][KS1][64]
  static constexpr int W2S = W1S + 2 * KS1 * 64;                 // [2][32][64]
  static constexpr int W2B = W2S + 2 * 32 * 64;                  // [2][32][64] (BWD only)
  static constexpr int B1P = W2B + (BWD ? 2 * 32 * 64 : 0);      // [2][32]
  static constexpr int B2P = B1P + 64;                           // [2][32]
  static constexpr int W3P = B2P + 64;                           // [AP][2][32]
  static constexpr int HC = W3P + AP * 64;                       // [8][4] head constants
  static constexpr int NORM = HC + 32;                           // mean[2*KS1], std[2*KS1]
  static constexpr int WAVE0 = (NORM + 4 * KS1 + 3) / 4 * 4;     // per-wave scratch
  static constexpr int T_FLOATS = 64 * TS;
  static constexpr int DO_FLOATS = 32 * 8;
  static constexpr int WAVE_FLOATS = BWD ? (2 * T_FLOATS + DO_FLOATS) : 0;
  static constexpr int TOTAL = WAVE0 + WAVES * WAVE_FLOATS;
  static constexpr int BYTES = TOTAL * 4;
};

// ---------------------------------------------------------------------------- staging

template <int KS1, int AP, bool BWD, bool ACTOR, int WAVES>
__device__ __forceinline__ void stage_weights(float* lds, const MlpArgs& a) {
  using L = Lds<KS1, AP, BWD, WAVES>;
  const int tid = threadIdx.x, nth = WAVES * 64;
  const int O = a.O, A = a.A;
  const float* W1 = a.params;
  const float* b1 = W1 + 64 * O;
  const float* W2 = b1 + 64;
  const float* b2 = W2 + 64 * 64;
  const float* tail = b2 + 64;
  const float* W3 = ACTOR ? tail + A : tail;
  const float* b3 = W3 + (ACTOR ? A * 64 : 64);
  // Global reads in memory order (coalesced); the permutation is applied on the LDS side.
  // Inverse of feat(): feature f sits at step q = (f>>5)<<4 | ((f>>3)&3)<<2 | (f&3), half (f>>2)&1.
  for (int idx = tid; idx < 2 * KS1 * 64; idx += nth) lds[L::W1S + idx] = 0.f;
  __syncthreads();
  for (int g = tid; g < 64 * O; g += nth) {
    const int row = g / O, k = g - row * O;
    const int t = row >> 5, i = row & 31, st = k >> 1, kh = k & 1;
    lds[L::W1S + (t * KS1 + st) * 64 + kh * 32 + i] = W1[g];
  }
  for (int g = tid; g < 64 * 64; g += nth) {
    const int row = g >> 6, col = g & 63;
    const float w = W2[g];
    {  // forward image: A operand row = output feature, k = input feature `col`
      const int t = row >> 5, i = row & 31;
      const int st = ((col >> 5) << 4) | (((col >> 3) & 3) << 2) | (col & 3), kh = (col >> 2) & 1;
      lds[L::W2S + (t * 32 + st) * 64 + kh * 32 + i] = w;
    }
    if (BWD) {  // backward image: A operand row = input feature `col`, k = output feature `row`
      const int t = col >> 5, i = col & 31;
      const int st = ((row >> 5) << 4) | (((row >> 3) & 3) << 2) | (row & 3), kh = (row >> 2) & 1;
      lds[L::W2B + (t * 32 + st) * 64 + kh * 32 + i] = w;
    }
  }
  for (int idx = tid; idx < 64; idx += nth) {
    const int h = idx >> 5, q = idx & 31;
    lds[L::B1P + idx] = b1[feat(q, h)];
    lds[L::B2P + idx] = b2[feat(q, h)];
  }
  for (int idx = tid; idx < AP * 64; idx += nth) {
    const int aa = idx >> 6, h = (idx >> 5) & 1, q = idx & 31;
    const int nout = ACTOR ? A : 1;
    lds[L::W3P + idx] = aa < nout ? W3[aa * 64 + feat(q, h)] : 0.f;
  }
  for (int idx = tid; idx < 8; idx += nth) {
    // head constants {bias, sigma, 1/(2 var), log sigma + log sqrt(2 pi)}
    float bias = 0.f, sigma = 1.f, half_inv_var = 0.f, logc = 0.f;
    if (ACTOR) {
      if (idx < A) {
        bias = b3[idx];
        const float ls = tail[idx];
        const float sp = ls > 20.f ? ls : log1pf(expf(ls));       // softplus, threshold 20
        sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);            // actors.py:63-64
        half_inv_var = 1.0f / (2.0f * (sigma * sigma));
        logc = logf(sigma) + kLogSqrt2Pi;
      }
    } else if (idx == 0) {
      bias = b3[0];
    }
    lds[L::HC + idx * 4 + 0] = bias;
    lds[L::HC + idx * 4 + 1] = sigma;
    lds[L::HC + idx * 4 + 2] = half_inv_var;
    lds[L::HC + idx * 4 + 3] = logc;
  }
  if (!ACTOR) {
    for (int idx = tid; idx < 2 * KS1; idx += nth) {
      lds[L::NORM + idx] = idx < O ? a.norm_mean[idx] : 0.f;
      lds[L::NORM + 2 * KS1 + idx] = idx < O ? a.norm_std[idx] : 1.f;
    }
  }
}

// --------------------------------------------------------------------- MFMA building blocks

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void load_bias(const float* bp, int h, f32x16& acc0, f32x16& acc1) {
  const f32x4* p = reinterpret_cast<const f32x4*>(bp + h * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v0 = p[j], v1 = p[4 + j];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc0[4 * j + c] = v0[c];
      acc1[4 * j + c] = v1[c];
    }
  }
}

// out[q] = tanh(bias + sum_k W[feat(q,h)][k] * in[k]) for this lane's sample (S layout).
template <int KS>
__device__ __forceinline__ void dense_tanh(const float* w_img, const float* bias_img,
                                           const float (&in)[KS], float (&out)[32], int lane) {
  f32x16 acc0, acc1;
  load_bias(bias_img, lane >> 5, acc0, acc1);
#pragma unroll
  for (int st = 0; st < KS; ++st) {
    const float a0 = w_img[st * 64 + lane];
    const float a1 = w_img[(KS + st) * 64 + lane];
    acc0 = mfma32(a0, in[st], acc0);
    acc1 = mfma32(a1, in[st], acc1);
  }
  // tanh in three batched sweeps (exp2, rcp, combine): 32 independent transcendental ops
  // back to back instead of 32 dependent exp->rcp chains, so one wave per SIMD still has ILP.
  float t[32];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    out[r] = acc0[r];
    out[16 + r] = acc1[r];
  }
#pragma unroll
  for (int r = 0; r < 32; ++r) t[r] = __builtin_amdgcn_exp2f(fabsf(out[r]) * -2.8853900817779268f);
  float d[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) d[r] = __builtin_amdgcn_rcpf(1.f + t[r]);
#pragma unroll
  for (int r = 0; r < 32; ++r) out[r] = copysignf((1.f - t[r]) * d[r], out[r]);
}

// S-layout register file -> per-wave scratch T[feature][sample].
__device__ __forceinline__ void scatter_S(float* T, const float (&v)[32], int s, int h) {
#pragma unroll
  for (int q = 0; q < 32; ++q) T[feat(q, h) * TS + s] = v[q];
}

// F layout: lane (c = lane&31, kh = lane>>5) gets feature 32*t + c, samples 16*kh .. 16*kh+15.
__device__ __forceinline__ void gather_F(const float* T, int t, int c, int kh, float (&out)[16]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(T + (32 * t + c) * TS + 16 * kh);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = p[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * j + e] = v[e];
  }
}

// Branch-free (see mlp64x16.hip): clamped addresses + select, never a load under a lane branch.
template <int KS1, bool NORMALISE>
__device__ __forceinline__ void load_obs(const MlpArgs& a, const float* norm, int64_t ns,
                                         bool valid, int h, float (&x)[KS1]) {
  const int64_t nc = valid ? ns : a.n - 1;
#pragma unroll
  for (int st = 0; st < KS1; ++st) {
    const int k = 2 * st + h;
    const int kc = k < a.O ? k : a.O - 1;
    float v = a.obs[nc * a.O + kc];
    if (NORMALISE)                                            // mean_stds.py:36-38
      v = __builtin_amdgcn_fmed3f((v - norm[kc]) / norm[2 * KS1 + kc], -a.norm_clip, a.norm_clip);
    x[st] = v * ((valid && k < a.O) ? 1.f : 0.f);   // mask-multiply: a select lets the load sink into a branch
  }
}

// Head pre-activation z[a] = b3[a] + sum_f W3[a][f] * h2[f] (both halves end with the sum).
template <int AP, int LW3P, int LHC>
__device__ __forceinline__ void head_linear(const float* lds, const float (&h2)[32], int h,
                                            float (&z)[AP]) {
  const f32x4* w3p = reinterpret_cast<const f32x4*>(lds + LW3P);
#pragma unroll
  for (int aa = 0; aa < AP; ++aa) {
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 w = w3p[(aa * 2 + h) * 8 + j];
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(h2[4 * j + e], w[e], part);
    }
    z[aa] = part + __shfl_xor(part, 32, 64) + lds[LHC + aa * 4];
  }
}

// ------------------------------------------------------------------------ forward kernels

// Acting: a2c.py:75-85.  One wave per 32 observations.
template <int KS1, int AP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ppo_act_kernel(MlpArgs a) {
  using L = Lds<KS1, AP, false, WAVES>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights<KS1, AP, false, true, WAVES>(lds, a);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31, h = lane >> 5;
  const int64_t ntiles = (a.n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles;
       tile += (int64_t)gridDim.x * WAVES) {
    const int64_t ns = tile * 32 + s;
    const bool valid = ns < a.n;
    float x[KS1], h1[32], h2[32], z[AP];
    load_obs<KS1, false>(a, nullptr, ns, valid, h, x);
    dense_tanh<KS1>(lds + L::W1S, lds + L::B1P, x, h1, lane);
    dense_tanh<32>(lds + L::W2S, lds + L::B2P, h1, h2, lane);
    head_linear<AP, L::W3P, L::HC>(lds, h2, h, z);
    float logp = 0.f;
#pragma unroll
    for (int aa = 0; aa < AP; ++aa) {
      if (aa < a.A) {
        const float loc = tanh_fast(z[aa]);
        const float sigma = lds[L::HC + aa * 4 + 1];
        float act = loc;
        if (a.eps != nullptr) act = loc + sigma * a.eps[(valid ? ns : a.n - 1) * a.A + aa];
        const float d = act - loc;
        logp += -(d * d) * lds[L::HC + aa * 4 + 2] - lds[L::HC + aa * 4 + 3];
        if (valid && h == 0) a.out0[ns * a.A + aa] = act;
      }
    }
    if (valid && h == 0 && a.out1 != nullptr) a.out1[ns] = logp;
  }
}

// One environment step of the on-policy collect loop in ONE launch (device-resident
// collectors): a2c.py:41-52 (act: forward + sample + log-prob) + a2c.py:58-69 (Segment.store of
// the whole transition + MeanStd.record).  Workgroups [0, act_blocks) run the policy on 32
// observations per wave and write actions / log-probs / observations straight into row `row`
// of the HBM-resident Segment; the LAST workgroup copies the transition outcome
// (next observations, rewards, resets, terminations) and replays MeanStd.record's sequential
// float32 accumulation from an LDS-staged copy of the observation block.
struct CollectArgs {
  MlpArgs act;                 // params, obs, eps, n = W, O, A; out0 = optional actions copy
  const float* next_obs; const float* rewards; const float* resets; const float* terminations;
  float* seg_obs; float* seg_act; float* seg_next; float* seg_rew; float* seg_rst;
  float* seg_term; float* seg_lp;
  float* norm_acc;
  int64_t row;
  int record_floats;           // LDS floats available to the record tile
};

template <int KS1, int AP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ppo_collect_kernel(CollectArgs c) {
  using L = Lds<KS1, AP, false, WAVES>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MlpArgs& a = c.act;
  const int64_t W = a.n;
  const int O = a.O, A = a.A;
  if (blockIdx.x == gridDim.x - 1) {
    // ---- transition outcome + MeanStd.record (segments.py:27-36, mean_stds.py:44-48)
    const int nth = WAVES * 64, tid = threadIdx.x;
    for (int64_t i = tid; i < W * O; i += nth) c.seg_next[c.row * W * O + i] = c.next_obs[i];
    for (int64_t i = tid; i < W; i += nth) {
      c.seg_rew[c.row * W + i] = c.rewards[i];
      c.seg_rst[c.row * W + i] = c.resets[i];
      c.seg_term[c.row * W + i] = c.terminations[i];
    }
    if (c.norm_acc == nullptr) return;
    float sum = 0.f, sum_sq = 0.f;
    if (tid < O) { sum = c.norm_acc[tid]; sum_sq = c.norm_acc[O + tid]; }
    const int64_t rows_per_chunk = c.record_floats / O;
    for (int64_t w0 = 0; w0 < W; w0 += rows_per_chunk) {
      const int64_t rows = min(rows_per_chunk, W - w0);
      __syncthreads();
      for (int64_t i = tid; i < rows * O; i += nth) lds[i] = a.obs[w0 * O + i];
      __syncthreads();
      if (tid < O) {
        record_rows(lds + tid, O, (int)rows, sum, sum_sq);
      }
    }
    if (tid < O) { c.norm_acc[tid] = sum; c.norm_acc[O + tid] = sum_sq; }
    return;
  }
  stage_weights<KS1, AP, false, true, WAVES>(lds, a);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31, h = lane >> 5;
  const int64_t ntiles = (W + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles;
       tile += (int64_t)(gridDim.x - 1) * WAVES) {
    const int64_t ns = tile * 32 + s;
    const bool valid = ns < W;
    float x[KS1], h1[32], h2[32], z[AP];
    load_obs<KS1, false>(a, nullptr, ns, valid, h, x);
    // this wave's 32 observation rows -> Segment row (contiguous block, coalesced)
    {
      const int64_t first = tile * 32 * O, count = min<int64_t>(32, W - tile * 32) * O;
      for (int64_t i = lane; i < count; i += 64)
        c.seg_obs[c.row * W * O + first + i] = a.obs[first + i];
    }
    dense_tanh<KS1>(lds + L::W1S, lds + L::B1P, x, h1, lane);
    dense_tanh<32>(lds + L::W2S, lds + L::B2P, h1, h2, lane);
    head_linear<AP, L::W3P, L::HC>(lds, h2, h, z);
    float logp = 0.f;
#pragma unroll
    for (int aa = 0; aa < AP; ++aa) {
      if (aa < A) {
        const float loc = tanh_fast(z[aa]);
        const float sigma = lds[L::HC + aa * 4 + 1];
        float act = loc;
        if (a.eps != nullptr) act = loc + sigma * a.eps[(valid ? ns : W - 1) * A + aa];
        const float d = act - loc;
        logp += -(d * d) * lds[L::HC + aa * 4 + 2] - lds[L::HC + aa * 4 + 3];
        if (valid && h == 0) {
          c.seg_act[(c.row * W + ns) * A + aa] = act;
          if (a.out0 != nullptr) a.out0[ns * A + aa] = act;
        }
      }
    }
    if (valid && h == 0) c.seg_lp[c.row * W + ns] = logp;
  }
}

// Critic forward: a2c.py:92-99.
template <int KS1, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void value_forward_kernel(MlpArgs a) {
  using L = Lds<KS1, 1, false, WAVES>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights<KS1, 1, false, false, WAVES>(lds, a);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31, h = lane >> 5;
  const int64_t ntiles = (a.n + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles;
       tile += (int64_t)gridDim.x * WAVES) {
    const int64_t ns = tile * 32 + s;
    const bool valid = ns < a.n;
    float x[KS1], h1[32], h2[32], z[1];
    load_obs<KS1, true>(a, lds + L::NORM, ns, valid, h, x);
    dense_tanh<KS1>(lds + L::W1S, lds + L::B1P, x, h1, lane);
    dense_tanh<32>(lds + L::W2S, lds + L::B2P, h1, h2, lane);
    head_linear<1, L::W3P, L::HC>(lds, h2, h, z);
    if (valid && h == 0) a.out0[ns] = z[0];
  }
}

// -------------------------------------------------------- fused forward + loss + backward

template <int KS1, int AP, bool ACTOR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void mlp64_grad_kernel(MlpArgs a) {
  using L = Lds<KS1, AP, true, WAVES>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (a.skip != nullptr && *a.skip != 0) return;
  stage_weights<KS1, AP, true, ACTOR, WAVES>(lds, a);
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, h = lane >> 5;
  const int c = s, kh = h;   // names used when the lane acts in F layout
  float* TA = lds + L::WAVE0 + wave * L::WAVE_FLOATS;
  float* TB = TA + L::T_FLOATS;
  float* DO = TB + L::T_FLOATS;
  const int O = a.O, A = a.A;

  float adv_mean = 0.f, adv_std = 1.f;
  bool adv_norm = false;
  if (ACTOR) {
    adv_mean = a.adv_stats[0];
    adv_std = a.adv_stats[1];
    adv_norm = a.adv_stats[3] != 0.f;
  }

  // accumulators that live across the whole tile loop
  f32x16 gW2[2][2], gW1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { gW2[i][0][r] = 0.f; gW2[i][1][r] = 0.f; gW1[i][r] = 0.f; }
  }
  float gW3[2][AP], gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gb3[AP], gsig[AP];
#pragma unroll
  for (int aa = 0; aa < AP; ++aa) { gW3[0][aa] = gW3[1][aa] = 0.f; gb3[aa] = 0.f; gsig[aa] = 0.f; }
  float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;   // loss, kl | sq err, v ; clipped ; n

  // Per-sample inputs of one tile, loaded one iteration ahead so that the global-load latency
  // hides under the previous tile's MFMA/VALU work (one wave per SIMD: nothing else would).
  struct TileIn {
    float x[KS1];
    float act[AP];
    float adv, old_lp, ret;
    bool valid;
  };
  auto load_tile = [&](int64_t tile, TileIn& in) {
    const int64_t ns = tile * 32 + s;
    in.valid = ns < a.n;
    load_obs<KS1, !ACTOR>(a, lds + L::NORM, ns, in.valid, h, in.x);
    const int64_t nc = in.valid ? ns : a.n - 1;
    in.adv = 0.f; in.old_lp = 0.f; in.ret = 0.f;
#pragma unroll
    for (int aa = 0; aa < AP; ++aa) in.act[aa] = 0.f;
    if (ACTOR) {
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) in.act[aa] = a.actions[nc * A + (aa < A ? aa : A - 1)];
      in.adv = a.adv[nc];
      in.old_lp = a.old_logp[nc];
    } else {
      in.ret = a.returns[nc];
    }
  };

  const int64_t ntiles = (a.n + 31) / 32;
  const int64_t tile_stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  TileIn cur, nxt;
  if (tile < ntiles) load_tile(tile, cur);
  for (; tile < ntiles; tile += tile_stride) {
    if (tile + tile_stride < ntiles) load_tile(tile + tile_stride, nxt);
    const bool valid = cur.valid;
    const bool counted = valid && h == 0;
    float h1[32], h2[32], z[AP], dzl[AP];
    float (&x)[KS1] = cur.x;
    dense_tanh<KS1>(lds + L::W1S, lds + L::B1P, x, h1, lane);
    dense_tanh<32>(lds + L::W2S, lds + L::B2P, h1, h2, lane);
    head_linear<AP, L::W3P, L::HC>(lds, h2, h, z);

    if (ACTOR) {
      // ---- ClippedRatio loss (updaters/actors.py:81-91) and d loss / d z
      float logp = 0.f, loc[AP], dif[AP];
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) {
        loc[aa] = 0.f; dif[aa] = 0.f;
        if (aa < A) {
          loc[aa] = tanh_fast(z[aa]);
          const float act = valid ? cur.act[aa] : loc[aa];
          dif[aa] = act - loc[aa];
          logp += -(dif[aa] * dif[aa]) * lds[L::HC + aa * 4 + 2] - lds[L::HC + aa * 4 + 3];
        }
      }
      const float old_lp = valid ? cur.old_lp : logp;
      float adv = valid ? cur.adv : 0.f;
      if (adv_norm) adv = (adv - adv_mean) / adv_std;             // segments.py:45
      const float ratio = expf(logp - old_lp);
      const float clipped_ratio = fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
      const float surr1 = adv * ratio, surr2 = adv * clipped_ratio;
      const bool outside = ratio > a.clip_hi || ratio < a.clip_lo;
      const bool dead = (ratio > a.clip_hi && adv > 0.f) || (ratio < a.clip_lo && adv < 0.f);
      const bool plain = a.plain != 0;                            // StochasticPolicyGradient
      const float g = (valid && (plain || !dead)) ? -(adv * (plain ? 1.f : ratio)) : 0.f;
      if (counted) {
        st0 += plain ? -(adv * logp) : -(surr2 < surr1 ? surr2 : surr1);   // torch.min: a NaN ratio stays NaN (fminf drops it)
        st1 += old_lp - logp;
        st2 += (outside && !plain) ? 1.f : 0.f;
        st3 += 1.f;
      }
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) {
        dzl[aa] = 0.f;
        if (aa < A) {
          const float sigma = lds[L::HC + aa * 4 + 1];
          const float inv_var = 2.f * lds[L::HC + aa * 4 + 2];
          const float dloc = g * dif[aa] * inv_var;                      // d logp / d loc
          dzl[aa] = dloc * (1.f - loc[aa] * loc[aa]);
          if (counted) {
            gsig[aa] += g * (dif[aa] * dif[aa] * inv_var / sigma - 1.f / sigma);
            gb3[aa] += dzl[aa];
          }
        }
      }
    } else {
      // ---- MSE (updaters/critics.py:20-21): d/dv of (v - ret)^2, unscaled by 1/N
      const float err = valid ? z[0] - cur.ret : 0.f;
      dzl[0] = 2.f * err;
      if (counted) {
        st0 += err * err;
        st1 += z[0];
        st3 += 1.f;
        gb3[0] += dzl[0];
      }
    }

    // ---- backward.  Two private LDS scratch tiles (TA, TB) so that every S->F transpose is
    // in flight while an independent MFMA chain or VALU block runs.
    scatter_S(TA, h2, s, h);                         // h2^T for dW3
    if (h == 0) {
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) {
        if (aa < 4) lo[aa] = dzl[aa]; else hi[aa - 4] = dzl[aa];
      }
      reinterpret_cast<f32x4*>(DO + s * 8)[0] = lo;
      if (AP > 4) reinterpret_cast<f32x4*>(DO + s * 8)[1] = hi;
    }
    // dz2 = (dzl . W3) * tanh'(h2), in place of h2 (its LDS copy is already issued)
    {
      const f32x4* w3p = reinterpret_cast<const f32x4*>(lds + L::W3P);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int aa = 0; aa < AP; ++aa) {
          const f32x4 w = w3p[(aa * 2 + h) * 8 + j];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(dzl[aa], w[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = h2[4 * j + e];
          h2[4 * j + e] = acc[e] * (1.f - y * y);
        }
      }
    }
    float (&dz2)[32] = h2;
    scatter_S(TB, dz2, s, h);                        // dz2^T for dW2 / db2
    wave_lds_sync();

    // dh1 = dz2 . W2 (transposed MFMA, stays in S layout) — covers the two transposes above
    f32x16 dacc0, dacc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dacc0[r] = 0.f; dacc1[r] = 0.f; }
    {
      const float* w2b = lds + L::W2B;
#pragma unroll
      for (int st = 0; st < 32; ++st) {
        dacc0 = mfma32(w2b[st * 64 + lane], dz2[st], dacc0);
        dacc1 = mfma32(w2b[(32 + st) * 64 + lane], dz2[st], dacc1);
      }
    }

    // dW3 (VALU, F layout) from TA / DO while the dh1 chain drains
    {
      float hF[2][16];
      gather_F(TA, 0, c, kh, hF[0]);
      gather_F(TA, 1, c, kh, hF[1]);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const f32x4 lo = reinterpret_cast<const f32x4*>(DO + (16 * kh + m) * 8)[0];
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        if (AP > 4) hi = reinterpret_cast<const f32x4*>(DO + (16 * kh + m) * 8)[1];
#pragma unroll
        for (int aa = 0; aa < AP; ++aa) {
          const float d = aa < 4 ? lo[aa] : hi[aa - 4];
          gW3[0][aa] = fmaf(hF[0][m], d, gW3[0][aa]);
          gW3[1][aa] = fmaf(hF[1][m], d, gW3[1][aa]);
        }
      }
    }
    float aF[2][16];
    gather_F(TB, 0, c, kh, aF[0]);
    gather_F(TB, 1, c, kh, aF[1]);
#pragma unroll
    for (int m = 0; m < 16; ++m) { gb2[0] += aF[0][m]; gb2[1] += aF[1][m]; }

    float dz1[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dz1[r] = dacc0[r] * (1.f - h1[r] * h1[r]);
      dz1[16 + r] = dacc1[r] * (1.f - h1[16 + r] * h1[16 + r]);
    }
    wave_lds_sync();                                 // TA / TB reads done
    scatter_S(TA, h1, s, h);
    scatter_S(TB, dz1, s, h);
    wave_lds_sync();

    // dW2[out][in] += dz2^T . h1 over the tile's 32 samples (MFMA, F-layout operands)
    {
      float bF[2][16];
      gather_F(TA, 0, c, kh, bF[0]);
      gather_F(TA, 1, c, kh, bF[1]);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        gW2[0][0] = mfma32(aF[0][m], bF[0][m], gW2[0][0]);
        gW2[0][1] = mfma32(aF[0][m], bF[1][m], gW2[0][1]);
        gW2[1][0] = mfma32(aF[1][m], bF[0][m], gW2[1][0]);
        gW2[1][1] = mfma32(aF[1][m], bF[1][m], gW2[1][1]);
      }
    }

    // dW1[out][in] += dz1^T . x
    {
      float cF[2][16], xF[16];
      gather_F(TB, 0, c, kh, cF[0]);
      gather_F(TB, 1, c, kh, cF[1]);
#pragma unroll
      for (int m = 0; m < 16; ++m) { gb1[0] += cF[0][m]; gb1[1] += cF[1][m]; }
      wave_lds_sync();                               // h1^T (TA) has been read
#pragma unroll
      for (int st = 0; st < KS1; ++st) TA[(2 * st + h) * TS + s] = x[st];
      wave_lds_sync();
      gather_F(TA, 0, c, kh, xF);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float xv = c < 2 * KS1 ? xF[m] : 0.f;     // rows >= 2*KS1 hold stale data
        gW1[0] = mfma32(cF[0][m], xv, gW1[0]);
        gW1[1] = mfma32(cF[1][m], xv, gW1[1]);
      }
      wave_lds_sync();                               // before the next tile rewrites TA / TB
    }
    cur = nxt;
  }

  // ---------------- fold the waves' accumulators into one image G (flat parameter layout)
  const int oW1 = 0, ob1 = 64 * O, oW2 = ob1 + 64, ob2 = oW2 + 4096, oTail = ob2 + 64;
  const int oLs = oTail, oW3 = ACTOR ? oTail + A : oTail, ob3 = oW3 + (ACTOR ? A * 64 : 64);
  const int P = ob3 + (ACTOR ? A : 1);
  float* G = lds + L::WAVE0;
  __syncthreads();
  for (int i = tid; i < P + kStatSlots; i += WAVES * 64) G[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h;
          G[oW2 + row * 64 + s] += gW2[ti][0][r];
          G[oW2 + row * 64 + 32 + s] += gW2[ti][1][r];
          if (s < O) G[oW1 + row * O + s] += gW1[ti][r];
        }
        const float vb1 = gb1[ti] + __shfl_xor(gb1[ti], 32, 64);
        const float vb2 = gb2[ti] + __shfl_xor(gb2[ti], 32, 64);
        if (kh == 0) { G[ob1 + 32 * ti + c] += vb1; G[ob2 + 32 * ti + c] += vb2; }
#pragma unroll
        for (int aa = 0; aa < AP; ++aa) {
          const float v = gW3[ti][aa] + __shfl_xor(gW3[ti][aa], 32, 64);
          if (kh == 0 && aa < (ACTOR ? A : 1)) G[oW3 + aa * 64 + 32 * ti + c] += v;
        }
      }
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) {
        const float vb = wave_sum(gb3[aa]);
        const float vs = wave_sum(gsig[aa]);
        if (lane == 0 && aa < (ACTOR ? A : 1)) {
          G[ob3 + aa] += vb;
          if (ACTOR) G[oLs + aa] += vs;     // d loss / d sigma; chain rule in the reducer
        }
      }
      const float r0 = wave_sum(st0), r1 = wave_sum(st1), r2 = wave_sum(st2), r3 = wave_sum(st3);
      if (lane == 0) { G[P + 0] += r0; G[P + 1] += r1; G[P + 2] += r2; G[P + 5] += r3; }
    }
    __syncthreads();
  }
  float* dst = a.out0 + (int64_t)blockIdx.x * a.pstride;
  for (int i = tid; i < P + kStatSlots; i += WAVES * 64) dst[i] = G[i];
}

// Fixed-order reduction of the per-workgroup partials (+ log_scale chain rule, entropy).
// One workgroup of 16 waves per 64 consecutive outputs: wave w sums a contiguous slice of the
// partial rows (independent loads, 8 in flight), the slices are combined in wave order.
constexpr int kReduceWaves = 16;

template <bool ACTOR>
__global__ __launch_bounds__(kReduceWaves * 64) void reduce_partials_kernel(
    const float* partials, int nblocks, int pstride, int P, const float* params,
    float* grad_sums, int oLs, int A, float entropy_coeff, double nloc, const int32_t* skip) {
  __shared__ double slices[kReduceWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const bool active = p < P + kStatSlots;
  if (skip != nullptr && *skip != 0) {
    // Skipped step: publish zeros, so a multi-GPU all-reduce of the (ignored) buffer stays finite.
    if (wave == 0 && active) grad_sums[p] = 0.f;
    return;
  }
  const int per = (nblocks + kReduceWaves - 1) / kReduceWaves;
  const int b0 = wave * per, b1 = min(b0 + per, nblocks);
  double acc = 0.0;
  if (active) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)(b + u) * pstride + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; b < b1; ++b) acc += (double)partials[(int64_t)b * pstride + p];
  }
  slices[wave][lane] = acc;
  __syncthreads();
  if (wave != 0 || !active) return;
  acc = 0.0;
#pragma unroll
  for (int w = 0; w < kReduceWaves; ++w) acc += slices[w][lane];
  float out = (float)acc;
  if (ACTOR) {
    if (p >= oLs && p < oLs + A) {
      const float ls = params[p];
      const float sp = ls > 20.f ? ls : log1pf(expf(ls));
      const float raw = sp + 1e-8f;
      const float sigma = fminf(fmaxf(raw, 1e-4f), 1.0f);
      const bool inside = raw >= 1e-4f && raw <= 1.0f;              // clamp passes the gradient
      const float dsig_dls = inside ? 1.f / (1.f + expf(-ls)) : 0.f; // softplus'
      double dsig = acc;
      if (entropy_coeff != 0.f) dsig -= (double)entropy_coeff / ((double)A * sigma) * nloc;
      out = (float)(dsig * dsig_dls);
    }
    if (p == P + 3 || p == P + 4) {
      // n-weighted entropy / std of the PRE-step distribution (actors.py:91,108)
      double ent = 0.0, sd = 0.0;
      for (int aa = 0; aa < A; ++aa) {
        const float ls = params[oLs + aa];
        const float sp = ls > 20.f ? ls : log1pf(expf(ls));
        const float sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);
        ent += (double)kEntropyConst + (double)logf(sigma);
        sd += sigma;
      }
      out = (float)((p == P + 3 ? ent : sd) / A * nloc);
    }
  }
  grad_sums[p] = out;
}

// ------------------------------------------------------------------------------- host side

namespace {

constexpr int kFwdWaves = 4;
constexpr int kMaxGradBlocks = 256;   // one workgroup per CU
// Waves per workgroup of the fused grad kernel: one wave per SIMD with the whole 512-register
// file (a two-waves-per-SIMD build spilled >100 registers and measured no faster).
std::atomic<int> g_grad_waves{4};
// 0 = 32x32x2 tiles, one wave per SIMD (mlp64_grad_kernel); 1 = 16x16x4 tiles, two waves per
// SIMD (mlp64x16.hip); 2 = 1 with the two 64x64 hidden-layer products of a tile on bf16x3 terms
// (six v_mfma_f32_16x16x32_bf16 per fp32 product, fp32 accumulation; Lds16<.., CH = 1>); 3 = 2 with
// dW2 on bf16x3 terms as well (2 x 2 tiles of v_mfma_f32_32x32x16_bf16; CH = 2); 4 = the three products on
// fp16x2 terms instead (three fp16 MFMAs per fp32 product, operands scaled into binary16's range; CH = 3).
constexpr int kDefaultGradVariant = 4;
std::atomic<int> g_grad_variant{kDefaultGradVariant};
// mlp64x16: wave priorities (0 off, 1 late half high, 2 the two waves of a SIMD swap priorities every tile: see the kernel)
std::atomic<int> g_grad_prio{2};
std::atomic<int> g_grad_skew{0};       // mlp64x16: optional start skew of waves 4-7, units of s_sleep(127); off:
                           // fp32 MFMA and VALU never overlap on gfx950, so there is no convoy to break

int ks1_bucket(int O) {
  if (O <= 4) return 2;
  if (O <= 18) return 9;
  if (O <= 32) return 16;
  return -1;
}
int ap_bucket(int A) {
  if (A <= 1) return 1;
  if (A <= 6) return 6;
  if (A <= 8) return 8;
  return -1;
}

int grad_blocks(int64_t n) {
  const int64_t tiles = (n + 31) / 32;
  int64_t blocks = (tiles + g_grad_waves - 1) / g_grad_waves;
  if (blocks > kMaxGradBlocks) blocks = kMaxGradBlocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <typename K>
int launch(K kernel, int blocks, int threads, int lds_bytes, hipStream_t stream, MlpArgs args,
           const char* what) {
  // >64 KiB of dynamic LDS needs an opt-in, once per kernel (not a stream operation, so it is
  // done outside any graph capture that may be active on later calls).
  static thread_local std::unordered_set<const void*> configured;
  const void* fn = reinterpret_cast<const void*>(kernel);
  if (configured.find(fn) == configured.end()) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute(%d B LDS): %s", what, lds_bytes, hipGetErrorString(e));
      return TONIC_ERR_LAUNCH;
    }
    configured.insert(fn);
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, args);
  TONIC_CHECK_LAUNCH(what);
  return TONIC_OK;
}

template <int KS1, int AP>
int launch_act(int blocks, hipStream_t st, const MlpArgs& a) {
  return launch(ppo_act_kernel<KS1, AP, kFwdWaves>, blocks, kFwdWaves * 64,
                Lds<KS1, AP, false, kFwdWaves>::BYTES, st, a, "tonic_ppo_act");
}
template <int KS1>
int launch_value(int blocks, hipStream_t st, const MlpArgs& a) {
  return launch(value_forward_kernel<KS1, kFwdWaves>, blocks, kFwdWaves * 64,
                Lds<KS1, 1, false, kFwdWaves>::BYTES, st, a, "tonic_value_forward");
}
#ifdef TONIC_DEV
template <int KS1, int AP, bool ACTOR>
int launch_grad(int blocks, hipStream_t st, const MlpArgs& a) {
  const char* what = ACTOR ? "tonic_ppo_actor_grad" : "tonic_value_regression_grad";
  return launch(mlp64_grad_kernel<KS1, AP, ACTOR, 4>, blocks, 256,
                Lds<KS1, AP, true, 4>::BYTES, st, a, what);
}
#endif

template <typename F>
int dispatch_ks1(int ks1, F&& f) {
  switch (ks1) {
    case 2: return f(std::integral_constant<int, 2>{});
    case 9: return f(std::integral_constant<int, 9>{});
    case 16: return f(std::integral_constant<int, 16>{});
    default: break;
  }
  set_error("unsupported observation bucket %d", ks1);
  return TONIC_ERR_UNSUPPORTED_SHAPE;
}

}  // namespace
}  // namespace tonic

using namespace tonic;

extern "C" int tonic_set_tuning(const char* key, int32_t value) {
  TONIC_REQUIRE(key != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_set_tuning: null key");
  if (strcmp(key, "grad_waves") == 0) {
    TONIC_REQUIRE(value == 4, TONIC_ERR_INVALID_ARGUMENT,
                  "grad_waves must be 4 (the only variant built), got %d", value);
    g_grad_waves = value;
    return TONIC_OK;
  }
  if (strcmp(key, "grad_skew") == 0) {
    TONIC_REQUIRE(value >= 0 && value <= 64, TONIC_ERR_INVALID_ARGUMENT,
                  "grad_skew must be in [0, 64], got %d", value);
    g_grad_skew = value;
    return TONIC_OK;
  }
  if (strcmp(key, "grad_prio") == 0) {
    TONIC_REQUIRE(value >= 0 && value <= 2, TONIC_ERR_INVALID_ARGUMENT,
                  "grad_prio must be 0, 1 or 2 (default), got %d", value);
    g_grad_prio = value;
    return TONIC_OK;
  }
  if (strcmp(key, "grad_variant") == 0) {
    TONIC_REQUIRE(value >= -1 && value <= 4, TONIC_ERR_INVALID_ARGUMENT,
                  "grad_variant must be 0 .. 4 or -1 (default), got %d", value);
#ifndef TONIC_DEV
    // the product library holds ONE form of the grad kernels; the fp32-MFMA / bf16x3 references the
    // parity tests compare it with live in libtonic_hip_dev.so (make dev)
    TONIC_REQUIRE(value < 0 || value == kDefaultGradVariant, TONIC_ERR_INVALID_ARGUMENT,
                  "grad_variant %d is a developer reference (libtonic_hip_dev.so); this library holds %d",
                  value, kDefaultGradVariant);
#endif
    g_grad_variant = value < 0 ? kDefaultGradVariant : value;
    return TONIC_OK;
  }
  if (strcmp(key, "policy_tail") == 0) {
    TONIC_REQUIRE(value == 0 || value == 1, TONIC_ERR_INVALID_ARGUMENT,
                  "policy_tail must be 0 or 1, got %d", value);
    g_policy_tail = value;
    return TONIC_OK;
  }
  if (strcmp(key, "chain_fault") == 0) {
    TONIC_REQUIRE(value == 0 || value == 1, TONIC_ERR_INVALID_ARGUMENT,
                  "chain_fault must be 0 or 1, got %d", value);
    g_chain_fault = value;
    return TONIC_OK;
  }
  if (strcmp(key, "q_chain") == 0) {
    TONIC_REQUIRE(value == 0 || value == 1, TONIC_ERR_INVALID_ARGUMENT,
                  "q_chain must be 0 or 1, got %d", value);
    g_q_chain = value;
    return TONIC_OK;
  }
  if (strcmp(key, "q_images") == 0) {      // 0: the off-policy passes on float32 MFMAs from the parameter blocks
    TONIC_REQUIRE(value == 0 || value == 1, TONIC_ERR_INVALID_ARGUMENT,
                  "q_images must be 0 or 1, got %d", value);
    g_q_images = value;
    return TONIC_OK;
  }
  if (strcmp(key, "gae_stream") == 0) {
    TONIC_REQUIRE(value >= 0 && value <= 4, TONIC_ERR_INVALID_ARGUMENT,
                  "gae_stream must be 0, 1, 2 (developer probe) or 3 (dword helpers), got %d", value);
    g_gae_stream = value;
    return TONIC_OK;
  }
  set_error("tonic_set_tuning: unknown key '%s'", key);
  return TONIC_ERR_INVALID_ARGUMENT;
}

extern "C" int tonic_get_tuning(const char* key, int32_t* value) {
  TONIC_REQUIRE(key != nullptr && value != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_get_tuning: null argument");
  if (strcmp(key, "grad_waves") == 0) { *value = g_grad_waves; return TONIC_OK; }
  if (strcmp(key, "grad_skew") == 0) { *value = g_grad_skew; return TONIC_OK; }
  if (strcmp(key, "grad_prio") == 0) { *value = g_grad_prio; return TONIC_OK; }
  if (strcmp(key, "grad_variant") == 0) { *value = g_grad_variant; return TONIC_OK; }
  if (strcmp(key, "policy_tail") == 0) { *value = g_policy_tail; return TONIC_OK; }
  if (strcmp(key, "gae_stream") == 0) { *value = g_gae_stream; return TONIC_OK; }
  if (strcmp(key, "q_chain") == 0) { *value = g_q_chain; return TONIC_OK; }
  if (strcmp(key, "q_images") == 0) { *value = g_q_images; return TONIC_OK; }
  set_error("tonic_get_tuning: unknown key '%s'", key);
  return TONIC_ERR_INVALID_ARGUMENT;
}

extern "C" int64_t tonic_ppo_actor_param_count(int32_t O, int32_t A) {
  return 64LL * O + 64 + 64 * 64 + 64 + A + 64LL * A + A;
}

extern "C" int64_t tonic_v_critic_param_count(int32_t O) {
  return 64LL * O + 64 + 64 * 64 + 64 + 64 + 1;
}

static int check_shape(int32_t O, int32_t A, bool actor) {
  TONIC_REQUIRE(O >= 1 && ks1_bucket(O) > 0, TONIC_ERR_UNSUPPORTED_SHAPE,
                "observation size %d: the fused kernels serve 1..32 (wider ones go through the "
                "entry points that take a workspace)", O);
  if (actor)
    TONIC_REQUIRE(A >= 1 && ap_bucket(A) > 0, TONIC_ERR_UNSUPPORTED_SHAPE,
                  "action size %d: the fused kernels serve 1..8 (more go through the entry points "
                  "that take a workspace)", A);
  return TONIC_OK;
}

static int check_wide(int32_t O, int32_t A, bool actor) {
  TONIC_REQUIRE(wide_supported(O, A, actor), TONIC_ERR_UNSUPPORTED_SHAPE,
                "observation size %d / action size %d: supported are O <= 384, A <= 32", O, A);
  return TONIC_OK;
}

extern "C" int64_t tonic_ppo_workspace_bytes(int64_t n, int32_t O, int32_t A, int32_t actor) {
  if (n <= 0 || O < 1 || (actor && A < 1)) return -1;
  if (wide_shape(O, A, actor != 0))
    return wide_supported(O, A, actor != 0) ? wide_workspace_bytes(n, O, A, actor != 0) : -1;
  const int64_t P = actor ? tonic_ppo_actor_param_count(O, A) : tonic_v_critic_param_count(O);
  return tonic_mlp64_grad_workspace_bytes(n, P);
}

extern "C" int tonic_ppo_act_wide(const float* d_actor_params, const float* d_observations,
                                  const float* d_eps, float* d_actions, float* d_log_probs,
                                  int64_t n, int32_t O, int32_t A, void* d_workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (!wide_shape(O, A, true))
    return tonic_ppo_act(d_actor_params, d_observations, d_eps, d_actions, d_log_probs, n, O, A,
                         stream);
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && n >= 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_act_wide: bad argument");
  if (int rc = check_wide(O, A, true)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.eps = d_eps;
  a.out0 = d_actions; a.out1 = d_log_probs; a.n = n; a.O = O; a.A = A;
  return wide_act(a, d_workspace, workspace_bytes, as_stream(stream));
}

extern "C" int tonic_value_forward_wide(const float* d_critic_params, const float* d_norm_mean,
                                        const float* d_norm_std, double norm_clip,
                                        const float* d_observations, float* d_values, int64_t n,
                                        int32_t O, void* d_workspace, int64_t workspace_bytes,
                                        void* stream) {
  if (!wide_shape(O, 1, false))
    return tonic_value_forward(d_critic_params, d_norm_mean, d_norm_std, norm_clip,
                               d_observations, d_values, n, O, stream);
  TONIC_REQUIRE(d_critic_params && d_norm_mean && d_norm_std && d_observations && d_values &&
                    n >= 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_value_forward_wide: bad argument");
  if (int rc = check_wide(O, 1, false)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_critic_params; a.obs = d_observations; a.norm_mean = d_norm_mean;
  a.norm_std = d_norm_std; a.norm_clip = clip_bound(norm_clip);
  a.out0 = d_values; a.n = n; a.O = O; a.A = 1;
  return wide_value(a, d_workspace, workspace_bytes, as_stream(stream));
}

extern "C" int tonic_ppo_act(const float* d_actor_params, const float* d_observations,
                             const float* d_eps, float* d_actions, float* d_log_probs,
                             int64_t n, int32_t O, int32_t A, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && n >= 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_act: null pointer or negative n");
  if (int rc = check_shape(O, A, true)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.eps = d_eps;
  a.out0 = d_actions; a.out1 = d_log_probs; a.n = n; a.O = O; a.A = A;
  const int64_t tiles = (n + 31) / 32;
  int blocks = (int)((tiles + kFwdWaves - 1) / kFwdWaves);
  if (blocks > 1024) blocks = 1024;
  const int ap = ap_bucket(A);
  hipStream_t st = as_stream(stream);
  return dispatch_ks1(ks1_bucket(O), [&](auto ks) {
    constexpr int KS1 = decltype(ks)::value;
    if (ap == 1) return launch_act<KS1, 1>(blocks, st, a);
    if (ap == 6) return launch_act<KS1, 6>(blocks, st, a);
    return launch_act<KS1, 8>(blocks, st, a);
  });
}

extern "C" int tonic_ppo_collect_step(
    const float* d_actor_params, const float* d_observations, const float* d_eps,
    const float* d_next_observations, const float* d_rewards, const float* d_resets,
    const float* d_terminations, float* d_seg_observations, float* d_seg_actions,
    float* d_seg_next_observations, float* d_seg_rewards, float* d_seg_resets,
    float* d_seg_terminations, float* d_seg_log_probs, float* d_norm_acc, float* d_actions_out,
    int64_t row, int64_t W, int32_t O, int32_t A, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_next_observations && d_rewards &&
                    d_resets && d_terminations && d_seg_observations && d_seg_actions &&
                    d_seg_next_observations && d_seg_rewards && d_seg_resets &&
                    d_seg_terminations && d_seg_log_probs,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_collect_step: null pointer");
  TONIC_REQUIRE(row >= 0 && W > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_ppo_collect_step: row=%lld W=%lld", (long long)row, (long long)W);
  if (int rc = check_shape(O, A, true)) return rc;
  CollectArgs c{};
  c.act.params = d_actor_params; c.act.obs = d_observations; c.act.eps = d_eps;
  c.act.out0 = d_actions_out; c.act.n = W; c.act.O = O; c.act.A = A;
  c.next_obs = d_next_observations; c.rewards = d_rewards; c.resets = d_resets;
  c.terminations = d_terminations; c.seg_obs = d_seg_observations; c.seg_act = d_seg_actions;
  c.seg_next = d_seg_next_observations; c.seg_rew = d_seg_rewards; c.seg_rst = d_seg_resets;
  c.seg_term = d_seg_terminations; c.seg_lp = d_seg_log_probs; c.norm_acc = d_norm_acc;
  c.row = row;
  const int64_t tiles = (W + 31) / 32;
  int act_blocks = (int)((tiles + kFwdWaves - 1) / kFwdWaves);
  if (act_blocks > 1024) act_blocks = 1024;
  const int ap = ap_bucket(A);
  hipStream_t st = as_stream(stream);
  return dispatch_ks1(ks1_bucket(O), [&](auto ks) -> int {
    constexpr int KS1 = decltype(ks)::value;
    auto go = [&](auto kernel, int act_lds) -> int {
      const int lds_bytes = act_lds > 65536 ? act_lds : 65536;
      c.record_floats = lds_bytes / 4;
      static thread_local const void* configured[16];
      static thread_local int n_configured = 0;
      const void* fn = reinterpret_cast<const void*>(kernel);
      bool known = false;
      for (int i = 0; i < n_configured; ++i) known |= configured[i] == fn;
      if (!known) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) {
          set_error("tonic_ppo_collect_step: hipFuncSetAttribute: %s", hipGetErrorString(e));
          return (int)TONIC_ERR_LAUNCH;
        }
        if (n_configured < 16) configured[n_configured++] = fn;
      }
      hipLaunchKernelGGL(kernel, dim3(act_blocks + 1), dim3(kFwdWaves * 64), lds_bytes, st, c);
      {
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
          set_error("tonic_ppo_collect_step: %s", hipGetErrorString(e2));
          return (int)TONIC_ERR_LAUNCH;
        }
      }
      return (int)TONIC_OK;
    };
    if (ap == 1) return go(ppo_collect_kernel<KS1, 1, kFwdWaves>, Lds<KS1, 1, false, kFwdWaves>::BYTES);
    if (ap == 6) return go(ppo_collect_kernel<KS1, 6, kFwdWaves>, Lds<KS1, 6, false, kFwdWaves>::BYTES);
    return go(ppo_collect_kernel<KS1, 8, kFwdWaves>, Lds<KS1, 8, false, kFwdWaves>::BYTES);
  });
}

constexpr int64_t kValues16MinRows = 32768;

extern "C" int tonic_value_forward(const float* d_critic_params, const float* d_norm_mean,
                                   const float* d_norm_std, double norm_clip,
                                   const float* d_observations,
                                   float* d_values, int64_t n, int32_t O, void* stream) {
  TONIC_REQUIRE(d_critic_params && d_norm_mean && d_norm_std && d_observations && d_values &&
                    n >= 0, TONIC_ERR_INVALID_ARGUMENT, "tonic_value_forward: bad argument");
  if (int rc = check_shape(O, 1, false)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_critic_params; a.obs = d_observations; a.norm_mean = d_norm_mean;
  a.norm_std = d_norm_std; a.norm_clip = clip_bound(norm_clip);
  a.out0 = d_values; a.n = n; a.O = O; a.A = 1;
  hipStream_t st = as_stream(stream);
  // A whole Segment (a2c.py:92-99 evaluates T x W observations twice per update): the forward half
  // of the regression kernel — 16-sample MFMA tiles with the inputs prefetched a tile ahead —
  // instead of the per-step kernel below, which is built around the latency of ONE step's rows.
  const int variant = g_grad_variant;
  if (n >= kValues16MinRows && variant >= 1 && grad16_supported(O, 1, false)) {
    a.out1 = d_values;
    return launch_values16(grad16_blocks(n), st, a, variant - 1);
  }
  const int64_t tiles = (n + 31) / 32;
  int blocks = (int)((tiles + kFwdWaves - 1) / kFwdWaves);
  if (blocks > 1024) blocks = 1024;
  return dispatch_ks1(ks1_bucket(O), [&](auto ks) {
    return launch_value<decltype(ks)::value>(blocks, st, a);
  });
}

extern "C" int64_t tonic_mlp64_grad_workspace_bytes(int64_t n, int64_t param_count) {
  const int64_t pstride = round_up(param_count + kStatSlots, 64);
  const int blocks = grad_blocks(n) > grad16_blocks(n) ? grad_blocks(n) : grad16_blocks(n);
  return (int64_t)blocks * pstride * (int64_t)sizeof(float);
}

template <bool ACTOR>
static int run_grad(MlpArgs a, int64_t P, float* d_grad_sums, float entropy_coeff,
                    int32_t max_workgroups, void* d_workspace, int64_t workspace_bytes,
                    void* stream) {
  TONIC_REQUIRE(max_workgroups >= 0, TONIC_ERR_INVALID_ARGUMENT,
                "max_workgroups must be 0 (one per compute unit) or positive, got %d",
                max_workgroups);
  const int variant = g_grad_variant;
  const bool use16 = variant >= 1 && grad16_supported(a.O, a.A, ACTOR);
  // The launch width is the CALLER's: a network whose iterations run under another kernel that
  // must keep its compute units (the PPO critic under the next rollout's resident collect kernel,
  // agents.py) leaves those units free.  The grouping of the float32 partial sums follows the
  // width, so a caller that wants reproducible bits passes the same width every time.
  const int all = use16 ? grad16_blocks(a.n) : grad_blocks(a.n);
  const int blocks = max_workgroups > 0 && max_workgroups < all ? max_workgroups : all;
  const int64_t pstride = round_up(P + kStatSlots, 64);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= blocks * pstride * (int64_t)sizeof(float),
                TONIC_ERR_WORKSPACE, "grad workspace too small: %lld < %lld",
                (long long)workspace_bytes, (long long)(blocks * pstride * sizeof(float)));
  a.out0 = static_cast<float*>(d_workspace);
  a.pstride = (int)pstride;
  a.skew = g_grad_skew;
  a.prio = g_grad_prio;
  const int ap = ACTOR ? ap_bucket(a.A) : 1;
  hipStream_t st = as_stream(stream);
#ifdef TONIC_DEV
  const int rc = use16 ? launch_grad16(ACTOR, blocks, st, a, variant - 1) : dispatch_ks1(ks1_bucket(a.O), [&](auto ks) {
    constexpr int KS1 = decltype(ks)::value;
    if constexpr (!ACTOR) {
      return launch_grad<KS1, 1, false>(blocks, st, a);
    } else {
      if (ap == 1) return launch_grad<KS1, 1, true>(blocks, st, a);
      if (ap == 6) return launch_grad<KS1, 6, true>(blocks, st, a);
      return launch_grad<KS1, 8, true>(blocks, st, a);
    }
  });
#else
  (void)ap;
  TONIC_REQUIRE(use16, TONIC_ERR_UNSUPPORTED_SHAPE, "fused grad kernel: O = %d, A = %d not served", a.O, a.A);
  const int rc = launch_grad16(ACTOR, blocks, st, a, variant - 1);
#endif
  if (rc != TONIC_OK) return rc;
  return launch_reduce_partials(ACTOR, static_cast<const float*>(d_workspace), blocks,
                                (int)pstride, (int)P, a.params, d_grad_sums, a.O, a.A,
                                entropy_coeff, (double)a.n, a.skip, st);
}

int tonic::launch_reduce_partials(bool actor, const float* partials, int blocks, int pstride, int P,
                                  const float* params, float* d_grad_sums, int O, int A,
                                  float entropy_coeff, double rows, const int32_t* skip,
                                  hipStream_t st, int log_scale_offset) {
  const int total = P + kStatSlots;
  const int oLs = log_scale_offset >= 0 ? log_scale_offset : 64 * O + 64 + 4096 + 64;
  const dim3 grid((total + 63) / 64), block(kReduceWaves * 64);
  if (actor)
    hipLaunchKernelGGL(reduce_partials_kernel<true>, grid, block, 0, st, partials, blocks, pstride,
                       P, params, d_grad_sums, oLs, A, entropy_coeff, rows, skip);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<false>, grid, block, 0, st, partials, blocks, pstride,
                       P, params, d_grad_sums, oLs, A, entropy_coeff, rows, skip);
  TONIC_CHECK_LAUNCH("reduce_partials_kernel");
  return TONIC_OK;
}

extern "C" int tonic_ppo_actor_grad(const float* d_actor_params, const float* d_observations,
                                    const float* d_actions, const float* d_advantages,
                                    const float* d_adv_stats, const float* d_old_log_probs,
                                    float* d_grad_sums, int64_t n, int32_t O, int32_t A,
                                    double ratio_clip, double entropy_coeff,
                                    const int32_t* d_skip_flag, int32_t max_workgroups,
                                    void* d_workspace, int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && d_advantages && d_adv_stats &&
                    d_old_log_probs && d_grad_sums && n > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_actor_grad: bad argument");
  const bool wide = wide_shape(O, A, true);
  if (int rc = wide ? check_wide(O, A, true) : check_shape(O, A, true)) return rc;
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.actions = d_actions;
  a.adv = d_advantages; a.adv_stats = d_adv_stats; a.old_logp = d_old_log_probs;
  a.skip = d_skip_flag; a.n = n; a.O = O; a.A = A;
  a.clip_lo = (float)(1.0 - ratio_clip);     // actors.py:85-86 (f64, then f32 in clamp)
  a.clip_hi = (float)(1.0 + ratio_clip);
  a.plain = ratio_clip < 0 ? 1 : 0;
  if (wide)
    return wide_actor_grad(a, d_grad_sums, (float)entropy_coeff, d_workspace, workspace_bytes,
                           as_stream(stream));
  return run_grad<true>(a, tonic_ppo_actor_param_count(O, A), d_grad_sums,
                        (float)entropy_coeff, max_workgroups,
                        d_workspace, workspace_bytes, stream);
}

// ---- any MLP(sizes, activation) torso: the layer-by-layer path (mlpwide.hip)
static int torso_from(int32_t layers, const int32_t* sizes, int32_t activation, Torso& t) {
  TONIC_REQUIRE(sizes != nullptr && layers >= 1 && layers <= kMaxTorsoLayers, TONIC_ERR_UNSUPPORTED_SHAPE,
                "torso: %d hidden layers (1 .. %d are served)", layers, kMaxTorsoLayers);
  t = Torso{layers, {0, 0, 0, 0}, activation};
  for (int l = 0; l < layers; ++l) t.size[l] = sizes[l];
  TONIC_REQUIRE(torso_supported(t), TONIC_ERR_UNSUPPORTED_SHAPE,
                "torso: layers of 4 .. 384 units (multiples of 4), activation 1 (Tanh) or 2 (ReLU)");
  return TONIC_OK;
}

extern "C" int64_t tonic_ppo_torso_param_count(int32_t O, int32_t A, int32_t actor, int32_t layers,
                                               const int32_t* sizes) {
  Torso t;
  if (torso_from(layers, sizes, 1, t) != TONIC_OK || !wide_supported(O, actor ? A : 1, actor != 0)) return -1;
  return torso_param_count(O, actor ? A : 1, actor != 0, t);
}

extern "C" int64_t tonic_ppo_torso_workspace_bytes(int64_t n, int32_t O, int32_t A, int32_t actor,
                                                   int32_t layers, const int32_t* sizes) {
  Torso t;
  if (n <= 0 || torso_from(layers, sizes, 1, t) != TONIC_OK ||
      !wide_supported(O, actor ? A : 1, actor != 0))
    return -1;
  return wide_workspace_bytes(n, O, actor ? A : 1, actor != 0, t);
}

extern "C" int tonic_ppo_act_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                                   const float* d_actor_params, const float* d_observations,
                                   const float* d_eps, float* d_actions, float* d_log_probs, int64_t n,
                                   int32_t O, int32_t A, void* d_workspace, int64_t workspace_bytes,
                                   void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && n >= 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_ppo_act_torso: null argument");
  Torso t;
  if (int rc = torso_from(layers, sizes, activation, t)) return rc;
  if (int rc = check_wide(O, A, true)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.eps = d_eps;
  a.out0 = d_actions; a.out1 = d_log_probs; a.n = n; a.O = O; a.A = A;
  return wide_act(a, d_workspace, workspace_bytes, as_stream(stream), t);
}

extern "C" int tonic_value_forward_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                                         const float* d_critic_params, const float* d_norm_mean,
                                         const float* d_norm_std, double norm_clip,
                                         const float* d_observations, float* d_values, int64_t n,
                                         int32_t O, void* d_workspace, int64_t workspace_bytes,
                                         void* stream) {
  TONIC_REQUIRE(d_critic_params && d_norm_mean && d_norm_std && d_observations && d_values && n >= 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_value_forward_torso: null argument");
  Torso t;
  if (int rc = torso_from(layers, sizes, activation, t)) return rc;
  if (int rc = check_wide(O, 1, false)) return rc;
  if (n == 0) return TONIC_OK;
  MlpArgs a{};
  a.params = d_critic_params; a.obs = d_observations; a.norm_mean = d_norm_mean;
  a.norm_std = d_norm_std; a.norm_clip = clip_bound(norm_clip);
  a.out0 = d_values; a.n = n; a.O = O; a.A = 1;
  return wide_value(a, d_workspace, workspace_bytes, as_stream(stream), t);
}

extern "C" int tonic_ppo_actor_grad_torso(int32_t layers, const int32_t* sizes, int32_t activation,
                                          const float* d_actor_params, const float* d_observations,
                                          const float* d_actions, const float* d_advantages,
                                          const float* d_adv_stats, const float* d_old_log_probs,
                                          float* d_grad_sums, int64_t n, int32_t O, int32_t A,
                                          double ratio_clip, double entropy_coeff,
                                          const int32_t* d_skip_flag, void* d_workspace,
                                          int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_observations && d_actions && d_advantages && d_adv_stats &&
                    d_old_log_probs && d_grad_sums && n > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_actor_grad_torso: bad argument");
  Torso t;
  if (int rc = torso_from(layers, sizes, activation, t)) return rc;
  if (int rc = check_wide(O, A, true)) return rc;
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.actions = d_actions;
  a.adv = d_advantages; a.adv_stats = d_adv_stats; a.old_logp = d_old_log_probs;
  a.skip = d_skip_flag; a.n = n; a.O = O; a.A = A;
  a.clip_lo = (float)(1.0 - ratio_clip);
  a.clip_hi = (float)(1.0 + ratio_clip);
  a.plain = ratio_clip < 0 ? 1 : 0;
  return wide_actor_grad(a, d_grad_sums, (float)entropy_coeff, d_workspace, workspace_bytes,
                         as_stream(stream), t);
}

extern "C" int tonic_value_regression_grad_torso(int32_t layers, const int32_t* sizes,
                                                 int32_t activation, const float* d_critic_params,
                                                 const float* d_norm_mean, const float* d_norm_std,
                                                 double norm_clip, const float* d_observations,
                                                 const float* d_returns, float* d_grad_sums, int64_t n,
                                                 int32_t O, void* d_workspace, int64_t workspace_bytes,
                                                 void* stream) {
  TONIC_REQUIRE(d_critic_params && d_norm_mean && d_norm_std && d_observations && d_returns &&
                    d_grad_sums && n > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_value_regression_grad_torso: bad argument");
  Torso t;
  if (int rc = torso_from(layers, sizes, activation, t)) return rc;
  if (int rc = check_wide(O, 1, false)) return rc;
  MlpArgs a{};
  a.params = d_critic_params; a.obs = d_observations; a.returns = d_returns;
  a.norm_mean = d_norm_mean; a.norm_std = d_norm_std; a.norm_clip = clip_bound(norm_clip);
  a.n = n; a.O = O; a.A = 1;
  return wide_critic_grad(a, d_grad_sums, d_workspace, workspace_bytes, as_stream(stream), t);
}

// Developer tool: per-phase s_memtime totals of the 8 waves of workgroup 0 of the 16x16x4 actor
// grad kernel (O <= 20, A <= 6 build).  d_phase_cycles: uint64[8 waves][12 phases].
extern "C" int tonic_debug_grad16_phases(const float* d_actor_params, const float* d_observations,
                                         const float* d_actions, const float* d_advantages,
                                         const float* d_adv_stats, const float* d_old_log_probs,
                                         int64_t n, int32_t O, int32_t A, void* d_workspace,
                                         int64_t workspace_bytes, uint64_t* d_phase_cycles,
                                         void* stream) {
  TONIC_REQUIRE(O == 17 && A == 6 && d_phase_cycles && d_workspace,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_debug_grad16_phases: unsupported shape");
  MlpArgs a{};
  a.params = d_actor_params; a.obs = d_observations; a.actions = d_actions;
  a.adv = d_advantages; a.adv_stats = d_adv_stats; a.old_logp = d_old_log_probs;
  a.n = n; a.O = O; a.A = A; a.clip_lo = 0.8f; a.clip_hi = 1.2f;
  a.out0 = static_cast<float*>(d_workspace);
  a.out1 = reinterpret_cast<float*>(d_phase_cycles);
  a.pstride = (int)round_up(tonic_ppo_actor_param_count(O, A) + kStatSlots, 64);
  a.skew = 0;
  TONIC_REQUIRE(workspace_bytes >= (int64_t)grad16_blocks(n) * a.pstride * 4, TONIC_ERR_WORKSPACE,
                "tonic_debug_grad16_phases: workspace too small");
  return launch_grad16_probe(grad16_blocks(n), as_stream(stream), a);
}

extern "C" int tonic_value_regression_grad(const float* d_critic_params,
                                           const float* d_norm_mean, const float* d_norm_std,
                                           double norm_clip,
                                           const float* d_observations, const float* d_returns,
                                           float* d_grad_sums, int64_t n, int32_t O,
                                           int32_t max_workgroups, void* d_workspace,
                                           int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_critic_params && d_norm_mean && d_norm_std && d_observations && d_returns &&
                    d_grad_sums && n > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_value_regression_grad: bad argument");
  const bool wide = wide_shape(O, 1, false);
  if (int rc = wide ? check_wide(O, 1, false) : check_shape(O, 1, false)) return rc;
  MlpArgs a{};
  a.params = d_critic_params; a.obs = d_observations; a.returns = d_returns;
  a.norm_mean = d_norm_mean; a.norm_std = d_norm_std; a.norm_clip = clip_bound(norm_clip);
  a.n = n; a.O = O; a.A = 1;
  if (wide) return wide_critic_grad(a, d_grad_sums, d_workspace, workspace_bytes, as_stream(stream));
  return run_grad<false>(a, tonic_v_critic_param_count(O), d_grad_sums, 0.f, max_workgroups,
                         d_workspace, workspace_bytes, stream);
}
