// gemm16: see gemm16.h.  gfx950, v_mfma_f32_16x16x4_f32, one wave per 16 x 32 output tile.
#include "gemm16.h"

namespace tonic {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four k-values (k = kbase + t) of one operand row/column `idx` for this lane.  Branch-free:
// `idx` and k are clamped to valid addresses by the caller / here, and whatever was read beyond K
// is zeroed by a 0/1 multiply at the consumer.  (A load under a lane-predicated branch gets its
// own `s_waitcnt vmcnt(0)`: three serialised memory round trips per k-chunk.)
template <bool KC>
__device__ __forceinline__ void load_operand(const float* __restrict__ P, int ld, int idx,
                                             int kbase, int K, bool vec, float (&v)[4]) {
  if (KC) {
    const float* p = P + (int64_t)idx * ld;
    if (vec) {                                     // wave-uniform: the chunk lies inside K
      // one global_load_dwordx4 per lane; gfx950 global loads only need dword alignment, so rows
      // of odd length (K = 111, 119: the first layers) take this path too
      typedef float f32x4_dword __attribute__((ext_vector_type(4), aligned(4)));
      const f32x4_dword q = *reinterpret_cast<const f32x4_dword*>(p + kbase);
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = p[min(kbase + t, K - 1)];
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = P[(int64_t)min(kbase + t, K - 1) * ld + idx];
  }
}

// One 16 x 32 output tile is owned by S = blockDim.y waves which interleave the 16-wide k-chunks
// (S = 1: no exchange; S up to 16 for the weight gradients, K = batch size) so that no wave walks
// more than a few chunks: the operands of up to four chunks are requested back to back and the
// MFMAs start when the first arrives.  Partial tiles are combined through LDS in wave order
// (deterministic).  blockDim = (64, S, tiles per workgroup).
constexpr int kGemmGroup = 4;                      // chunks in flight per wave
constexpr int kGemmMaxWaves = 16;

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm16_tiles(const GemmArgs& g, int block_index, float* part) {
  const int lane = threadIdx.x, part_of = threadIdx.y, S = blockDim.y;
  const int i = lane & 15, kg = lane >> 4;
  const int tiles_n = (g.N + 31) / 32, tiles_m = (g.M + 15) / 16;
  const int tile = block_index * blockDim.z + threadIdx.z;
  const bool live = tile < tiles_m * tiles_n;      // uniform per wave
  const int tm = live ? tile / tiles_n : 0, tn = live ? tile - tm * tiles_n : 0;
  const int m0 = tm * 16, n0 = tn * 32;
  const int z = blockIdx.z;
  const float* A = g.A + z * g.strideA;
  const float* B = g.B + z * g.strideB;
  float* C = g.C + z * g.strideC;
  const bool vec_a = A_KC, vec_b = B_KC;
  const int ia = min(m0 + i, g.M - 1), ib0 = min(n0 + i, g.N - 1), ib1 = min(n0 + 16 + i, g.N - 1);

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float colsum = 0.f;
  const int chunks = (g.K + 15) / 16;
  for (int first = part_of; live && first < chunks; first += kGemmGroup * S) {
    float a[kGemmGroup][4], b0[kGemmGroup][4], b1[kGemmGroup][4];
#pragma unroll
    for (int j = 0; j < kGemmGroup; ++j) {         // all requests first (clamped, never skipped)
      const int k0 = 16 * min(first + j * S, chunks - 1);
      const bool inside = k0 + 16 <= g.K;
      load_operand<A_KC>(A, g.lda, ia, k0 + 4 * kg, g.K, vec_a && inside, a[j]);
      load_operand<B_KC>(B, g.ldb, ib0, k0 + 4 * kg, g.K, vec_b && inside, b0[j]);
      load_operand<B_KC>(B, g.ldb, ib1, k0 + 4 * kg, g.K, vec_b && inside, b1[j]);
    }
#pragma unroll
    for (int j = 0; j < kGemmGroup; ++j) {
      const int c = first + j * S;
      if (c >= chunks) break;                      // wave-uniform
      const int kbase = 16 * c + 4 * kg;
      if (16 * c + 16 > g.K) {                     // ragged last chunk: zero what lies beyond K
#pragma unroll
        for (int t = 0; t < 4; ++t) a[j][t] *= (kbase + t < g.K) ? 1.f : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma16(a[j][t], b0[j][t], acc0);
        acc1 = mfma16(a[j][t], b1[j][t], acc1);
      }
      if (!A_KC) colsum += (a[j][0] + a[j][1]) + (a[j][2] + a[j][3]);
    }
  }

  if (S > 1) {
    float* mine = part + ((threadIdx.z * S + part_of) * 64 + lane) * 9;
#pragma unroll
    for (int r = 0; r < 4; ++r) { mine[r] = acc0[r]; mine[4 + r] = acc1[r]; }
    mine[8] = colsum;
    __syncthreads();
    if (part_of != 0 || !live) return;
    for (int w = 1; w < S; ++w) {                  // fixed order: bit-reproducible
      const float* other = part + ((threadIdx.z * S + w) * 64 + lane) * 9;
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc0[r] += other[r]; acc1[r] += other[4 + r]; }
      colsum += other[8];
    }
  } else if (!live) {
    return;
  }

  if (!A_KC && g.colsum != nullptr && tn == 0) {
    colsum += __shfl_xor(colsum, 16, 64);
    colsum += __shfl_xor(colsum, 32, 64);
    if (kg == 0 && m0 + i < g.M) g.colsum[z * g.strideColsum + m0 + i] = colsum;
  }

  const float* bias = g.bias ? g.bias + z * g.strideBias : nullptr;
  const float* mask = g.mask ? g.mask + z * g.strideMask : nullptr;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = n0 + 16 * half + i;
    if (n >= g.N) continue;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * kg + r;
      if (m >= g.M) continue;
      float v = (half == 0 ? acc0[r] : acc1[r]) * g.alpha + bn;
      if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (g.act == ACT_TANH) v = tanhf(v);
      else if (g.act == ACT_ELU) v = v > 0.f ? v : expm1f(v);       // torch.nn.ELU, alpha = 1
      if (mask != nullptr) {
        const float a = mask[(int64_t)m * g.ldmask + n];
        if (g.mask_act == ACT_TANH) v = v * (1.f - a * a);
        else if (g.mask_act == ACT_ELU) v = a > 0.f ? v : v * (a + 1.f);
        else if (!(a > 0.f)) v = 0.f;
      }
      float* dst = C + (int64_t)m * g.ldc + n;
      *dst = g.accumulate ? *dst + v : v;
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * kGemmMaxWaves) void gemm16_kernel(GemmArgs g) {
  extern __shared__ float part[];                  // [waves][64][9] when S > 1
  gemm16_tiles<A_KC, B_KC>(g, blockIdx.x, part);
}

// Several independent GEMMs of the same operand layout and contraction length in ONE launch (the
// weight gradients of one network: dW1, dW2, dW3 ... all contract over the batch): workgroup
// ranges [first[p], first[p + 1]) belong to problem p.  A launch costs ~5 us whatever it computes.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(64 * kGemmMaxWaves) void gemm16_group_kernel(GemmGroup G) {
  extern __shared__ float part[];
  int p = 0;
#pragma unroll
  for (int q = 1; q < kGemmGroupMax; ++q) p += (q < G.count && (int)blockIdx.x >= G.first[q]) ? 1 : 0;
  gemm16_tiles<A_KC, B_KC>(G.problem[p], (int)blockIdx.x - G.first[p], part);
}

// ---------------------------------------------------------------------------------------------
// Weight gradients: C[M,N] = A[K,M]^T . B[K,N] with K = batch size (TN), bias gradient = column
// sums of A.  Both operands are k-strided, i.e. contiguous ALONG the output index, so one lane can
// fetch two neighbouring output columns with one 8-byte load: a wave owns a 32 x 32 output tile
// made of 2 x 2 INTERLEAVED 16 x 16 MFMA tiles (tile jm holds rows m0 + 2 i + jm, tile jn columns
// n0 + 2 i + jn), 8 loads feed 16 MFMAs per 16-k chunk (the generic kernel above: 12 dword loads
// and ~70 address instructions per 8 MFMAs).  Four waves share a tile and interleave the chunks;
// each keeps kTnDepth chunks in flight.  Partial tiles and column sums are folded through LDS in
// wave order.
constexpr int kTnWaves = 4, kTnDepth = 4, kTnPart = 18;      // floats per lane in the exchange

typedef float f32x2_dword __attribute__((ext_vector_type(2), aligned(4)));

// Two neighbouring columns of one operand row.  `base` is wave-uniform (the chunk's first row),
// `offset` the lane's UNSIGNED 32-bit element offset from it (scalar base + zero-extended vector
// offset is an addressing mode of global_load): the load needs no vector address arithmetic.
// EDGE (compile-time: a run-time flag, even a wave-uniform one, puts a branch around every
// load): tiles that reach past the last row / column of C take clamped scalar loads.
// Inside the main loop the non-EDGE form is a BUFFER load (tn_buffer below): the operand as a
// buffer resource, the lane's byte offset, the chunk's byte offset as the scalar operand — the
// compiler hoists the zero-extension of `offset` out of the loop and then adds scalar base and
// 64-bit vector offset with a v_lshl_add_u64 per load, which fp32 MFMAs do not hide.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tn_buffer(const float* tensor) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tensor), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void tn_buffer_load(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes,
                                               unsigned chunk_bytes, float (&v)[2]) {
  const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, chunk_bytes, 0);
  v[0] = __uint_as_float(q[0]); v[1] = __uint_as_float(q[1]);
}

template <bool EDGE>
__device__ __forceinline__ void tn_load(const float* __restrict__ base, unsigned offset, int col,
                                        int cols, float (&v)[2]) {
  if (!EDGE) {                              // (byte offset: the addressing mode takes bytes)
    const f32x2_dword q = *reinterpret_cast<const f32x2_dword*>(
        reinterpret_cast<const char*>(base) + 4u * offset);
    v[0] = q[0]; v[1] = q[1];
  } else {                                  // offset = row offset + col: re-clamp the two columns
    const float* row = base + (offset - col);
    v[0] = row[min(col, cols - 1)];
    v[1] = row[min(col + 1, cols - 1)];
  }
}

// The step's constants (adam.py:530-547: float64 bias corrections, then rounded) — the expressions
// of adam_kernel.
struct AdamConsts { float step_size, bias2_sqrt, w1, w2; };

__device__ __forceinline__ AdamConsts adam_consts(const AdamFold& f) {
  if (f.consts != nullptr) {          // formed by the host in float64 like the reference's Python
    AdamConsts c;                     // floats (adam.py:530-536): no float64 pow on the device
    c.step_size = f.consts[0]; c.bias2_sqrt = f.consts[1];
    c.w1 = (float)(1.0 - f.beta1_d); c.w2 = (float)(1.0 - f.beta2_d);
    return c;
  }
  const int step = f.state[0] + 1;
  const double bias1 = 1.0 - pow(f.beta1_d, (double)step);
  const double bias2 = 1.0 - pow(f.beta2_d, (double)step);
  AdamConsts c;
  c.step_size = (float)(f.lr_d / bias1);
  c.bias2_sqrt = (float)sqrt(bias2);
  c.w1 = (float)(1.0 - f.beta1_d); c.w2 = (float)(1.0 - f.beta2_d);
  return c;
}

// One element: the gradient SUM just formed -> new parameter (moments updated in place).
__device__ __forceinline__ float adam_element(float sum, float p, float& m, float& v,
                                              const AdamFold& f, const AdamConsts& c) {
  const float gr = sum * f.grad_scale;
  m = m + c.w1 * (gr - m);                                         // lerp_, adam.py:457
  v = v * f.beta2 + c.w2 * (gr * gr);                              // mul_().addcmul_(), :476
  const float denom = sqrtf(v) / c.bias2_sqrt + f.eps;             // :545
  return p - c.step_size * (m / denom);                            // addcdiv_, :547
}

__device__ __forceinline__ void adam_apply(const AdamFold& f, const AdamConsts& c, int64_t off,
                                           float sum) {
  float m = f.exp_avg[off], v = f.exp_avg_sq[off];
  const float p = adam_element(sum, f.params[off], m, v, f, c);
  f.params[off] = p; f.exp_avg[off] = m; f.exp_avg_sq[off] = v;
  if (f.target != nullptr) f.target[off] = f.target[off] * f.polyak_keep + f.polyak_mix * p;
}

// The last workgroup of the launch: step counter and logged statistics (adam_finalize, optim.hip).
__device__ __forceinline__ void adam_fold_arrive(const AdamFold& f, unsigned total, bool stepping) {
  unsigned* arrivals = reinterpret_cast<unsigned*>(f.state + 3);
  // (this wave's loads of the step's state / constants have returned before it arrives: the
  //  finaliser's write of the step counter races with nobody, see adam_kernel in optim.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  if (before != total - 1) return;
  __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!stepping) {                            // not a training step: see AdamFold::skip
    if (f.info_row != nullptr) {              // {loss = NaN, ..., ran = 0, gave up = 1}
      f.info_row[0] = __builtin_nanf(""); f.info_row[6] = 0.f; f.info_row[7] = 1.f;
    }
    return;
  }
  f.state[0] += 1;
  if (f.info_row == nullptr) return;
  const float* st = f.grads + f.n;
  if (f.stats_kind == 3) {
    f.info_row[0] = st[0] * f.grad_scale;      // loss_1 + loss_2 (critics.py:172,224)
    f.info_row[1] = st[1] * f.grad_scale;      // mean q1
    f.info_row[2] = st[2] * f.grad_scale;      // mean q2
    f.info_row[6] = 1.f;
  } else if (f.stats_kind == 4) {
    f.info_row[0] = st[0] * f.grad_scale;      // actor loss (actors.py:179,257)
    f.info_row[6] = 1.f;
  }
}

// COLSUM: this tile also forms the column sums of A (the bias gradient: the tiles of the first tile
// column) — a compile-time flag, or the six additions per chunk run (selected away) in every tile
template <bool EDGE, bool COLSUM>
__device__ __forceinline__ void gemm_tn_tile(const GemmArgs& g, int tm, int tn, float* part,
                                             const AdamFold& fold, unsigned total_blocks,
                                             unsigned long long* stamps) {
  const bool probe = stamps != nullptr && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0 &&
                     threadIdx.y == 0;
  auto stamp = [&](int k) {
    if (probe) {
      __builtin_amdgcn_sched_barrier(0);
      stamps[k] = wall_clock64();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  const int lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);       // scalar: uniform control flow
  const int i = lane & 15, kg = lane >> 4;
  const int m0 = 32 * tm, n0 = 32 * tn;
  const int z = blockIdx.z;
  const float* A = g.A + z * g.strideA;
  const float* B = g.B + z * g.strideB;
  float* C = g.C + z * g.strideC;
  const int ca = m0 + 2 * i, cb = n0 + 2 * i;
  const int full = g.K / 16;                // full 16-k chunks; a ragged one may follow
  const int mine = full > w ? (full - w + kTnWaves - 1) / kTnWaves : 0;       // chunks w, w+4, ...
  unsigned offa[4], offb[4];                // rows 4 kg + t of a chunk
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    offa[t] = (4 * kg + t) * g.lda + ca;
    offb[t] = (4 * kg + t) * g.ldb + cb;
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int jm = 0; jm < 2; ++jm)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) acc[jm][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
  float colsum[2] = {0.f, 0.f};
  // (the bias gradients = column sums of A are the business of the tiles of the first tile column)
  constexpr bool want_colsum = COLSUM;
  const __amdgpu_buffer_rsrc_t bufA = tn_buffer(A), bufB = tn_buffer(B);

  auto multiply = [&](const float (&a)[4][2], const float (&b)[4][2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int jm = 0; jm < 2; ++jm) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[jm][jn] = mfma16(a[t][jm], b[t][jn], acc[jm][jn]);
      }
    }
    if (want_colsum) {
#pragma unroll
      for (int jm = 0; jm < 2; ++jm) colsum[jm] += (a[0][jm] + a[1][jm]) + (a[2][jm] + a[3][jm]);
    }
  };

  // ---- full chunks: kTnDepth chunks in flight in rotating register slots; the loads that
  //      refill a slot are issued right after the MFMAs that drained it, so that load issue
  //      (~7 ns of the CU's memory pipe per instruction, four waves sharing it) and MFMA issue
  //      interleave instead of alternating in bursts (profiles/r01_ubench_row_stride.md)
  float a[kTnDepth][4][2], b[kTnDepth][4][2];
  auto request = [&](int slot, int index) {
    const int c = min(w + kTnWaves * index, full - 1);           // past the end: re-read, unused
    if (!EDGE) {
      const unsigned sa = 64u * (unsigned)(c * g.lda), sb = 64u * (unsigned)(c * g.ldb);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tn_buffer_load(bufA, 4u * offa[t], sa, a[slot][t]);
        tn_buffer_load(bufB, 4u * offb[t], sb, b[slot][t]);
      }
    } else {
      const float* Ac = A + (int64_t)16 * c * g.lda;             // scalar arithmetic
      const float* Bc = B + (int64_t)16 * c * g.ldb;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tn_load<EDGE>(Ac, offa[t], ca, g.M, a[slot][t]);
        tn_load<EDGE>(Bc, offb[t], cb, g.N, b[slot][t]);
      }
    }
  };
  if (mine > 0) {
#pragma unroll
    for (int d = 0; d < kTnDepth; ++d) request(d, d);
  }
  // optimizer epilogue: the step's constants (two float64 pow: ~1 us of VALU work) are formed HERE,
  // by every wave, while the first operand chunks are in flight
  AdamConsts consts{};
  bool stepping = fold.on != 0;               // (uniform)
  if (fold.on) {
    __builtin_amdgcn_sched_barrier(0);
    consts = adam_consts(fold);
    if (fold.skip != nullptr)
      stepping = __hip_atomic_load(fold.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    __builtin_amdgcn_sched_barrier(0);
  }
  stamp(1);                                         // first requests out, constants formed
  for (int first = 0; first < mine; first += kTnDepth) {
#pragma unroll
    for (int d = 0; d < kTnDepth; ++d) {
      if (first + d < mine) multiply(a[d], b[d]);                // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
      request(d, first + d + kTnDepth);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the ragged last chunk (K % 16 rows) belongs to the wave whose turn it is
  if (g.K % 16 != 0 && full % kTnWaves == w) {
    const float* Ac = A + (int64_t)16 * full * g.lda;
    const float* Bc = B + (int64_t)16 * full * g.ldb;
    const int last = g.K - 1 - 16 * full;
    float a[4][2], b[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = min(4 * kg + t, last);                       // clamp the row, zero it below
      tn_load<EDGE>(Ac, (unsigned)(r * g.lda + ca), ca, g.M, a[t]);
      tn_load<EDGE>(Bc, (unsigned)(r * g.ldb + cb), cb, g.N, b[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool keep = 4 * kg + t <= last;
      a[t][0] = keep ? a[t][0] : 0.f; a[t][1] = keep ? a[t][1] : 0.f;
    }
    multiply(a, b);
  }

  stamp(2);                                         // main loop done
  float* slot = part + (w * 64 + lane) * kTnPart;
#pragma unroll
  for (int jm = 0; jm < 2; ++jm)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[(2 * jm + jn) * 4 + r] = acc[jm][jn][r];
  slot[16] = colsum[0]; slot[17] = colsum[1];
  __syncthreads();
  stamp(3);                                         // partials exchanged
  // Epilogue, spread over the four waves: wave w finishes register row r = w of the four MFMA tiles
  // (rows m0 + 2 (4 kg + w) + {0, 1}, columns cb, cb + 1).  Every wave adds the four partials in
  // wave order — ((p0 + p1) + p2) + p3, bit-reproducible — so the 32 x 32 tile's stores and the
  // optimizer arithmetic (~60 instructions per element) are not left to one wave of four.
  const int64_t goff = fold.on ? (g.C + z * g.strideC) - fold.grads : 0;      // tile base in the block
  float fin[2][2];
#pragma unroll
  for (int jm = 0; jm < 2; ++jm) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      float v = part[lane * kTnPart + (2 * jm + jn) * 4 + w];
#pragma unroll
      for (int o = 1; o < kTnWaves; ++o) v += part[(o * 64 + lane) * kTnPart + (2 * jm + jn) * 4 + w];
      fin[jm][jn] = v;
    }
  }
  if (w == 0) {
    if (fold.on && lane == 0) adam_fold_arrive(fold, total_blocks, stepping);   // (state / constants were read)
    if (g.colsum != nullptr && tn == 0) {
#pragma unroll
      for (int jm = 0; jm < 2; ++jm) {
        float v = colsum[jm];
        for (int o = 1; o < kTnWaves; ++o) v += part[(o * 64 + lane) * kTnPart + 16 + jm];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kg == 0 && ca + jm < g.M) {
          g.colsum[z * g.strideColsum + ca + jm] = v;
          if (stepping)
            adam_apply(fold, consts, (g.colsum + z * g.strideColsum + ca + jm) - fold.grads, v);
        }
      }
    }
  }
  // optimizer epilogue: parameter / moment / target loads of this wave's two rows first
  const bool polyak = fold.on && fold.target != nullptr;
  float pm[2][4][2];
  if (fold.on) {
#pragma unroll
    for (int jm = 0; jm < 2; ++jm) {
      const int m = min(m0 + 2 * (4 * kg + w) + jm, g.M - 1);
      const int64_t off = goff + (int64_t)m * g.ldc + min(cb, g.N - 1);
      const float* src[4] = {fold.params + off, fold.exp_avg + off, fold.exp_avg_sq + off,
                             (polyak ? fold.target : fold.params) + off};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!EDGE) {
          const f32x2_dword v = *reinterpret_cast<const f32x2_dword*>(src[q]);
          pm[jm][q][0] = v[0]; pm[jm][q][1] = v[1];
        } else {
          pm[jm][q][0] = src[q][0];
          pm[jm][q][1] = cb + 1 < g.N ? src[q][1] : 0.f;
        }
      }
    }
  }
  stamp(4);                                         // folded, optimizer operands requested
  // D layout: lane (column index i, group kg), register r <-> row index 4 kg + r of the MFMA tile
  float newp[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, newt[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // -> the weight images
#pragma unroll
  for (int jm = 0; jm < 2; ++jm) {
    const int m = m0 + 2 * (4 * kg + w) + jm;
    if (m >= g.M) continue;
    float* dst = C + (int64_t)m * g.ldc + cb;
    float v0 = fin[jm][0] * g.alpha, v1 = fin[jm][1] * g.alpha;
    if (cb + 1 < g.N) {
      if (g.accumulate) {
        const f32x2_dword old = *reinterpret_cast<const f32x2_dword*>(dst);
        v0 += old[0]; v1 += old[1];
      }
      *reinterpret_cast<f32x2_dword*>(dst) = f32x2_dword{v0, v1};
    } else if (cb < g.N) {
      v0 = g.accumulate ? dst[0] + v0 : v0;
      dst[0] = v0;
    }
    if (!stepping) continue;
    // ---- Adam (+ polyak) on the two elements just formed
    const float sum[2] = {v0, v1};
    const int64_t off = goff + (int64_t)m * g.ldc + cb;
    float out[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float mo = pm[jm][1][e], vo = pm[jm][2][e];
      const float p = adam_element(sum[e], pm[jm][0][e], mo, vo, fold, consts);
      out[0][e] = p; out[1][e] = mo; out[2][e] = vo;
      out[3][e] = pm[jm][3][e] * fold.polyak_keep + fold.polyak_mix * p;
      newp[jm][e] = p; newt[jm][e] = out[3][e];
    }
    float* to[4] = {fold.params + off, fold.exp_avg + off, fold.exp_avg_sq + off, fold.target + off};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q == 3 && !polyak) continue;
      if (cb + 1 < g.N) {
        *reinterpret_cast<f32x2_dword*>(to[q]) = f32x2_dword{out[q][0], out[q][1]};
      } else if (cb < g.N) {
        to[q][0] = out[q][0];
      }
    }
  }
  // ---- the tensor's fp16x2 weight images (mlpimg.h) follow the parameters: the two rows x two columns this
  //      lane has just stepped — the forward image pairs them along the columns (k), the transposed one along
  //      the rows; what lies beyond the tensor stays zero.  Same conversion as build_images_kernel -> same bits.
  const ImgTarget& im = g.img;
  if (stepping && (im.fwd != nullptr || im.bwd != nullptr)) {       // uniform
    const int r_even = m0 + 2 * (4 * kg + w);
    const int64_t zoff = z * im.stride;
    if (im.fwd != nullptr && cb < g.N) {
#pragma unroll
      for (int jm = 0; jm < 2; ++jm) {
        if (r_even + jm >= g.M) continue;
        const bool second = cb + 1 < g.N;
        img_store_pair(im.fwd + zoff, im.fwd_chunks, r_even + jm, cb, newp[jm][0], second ? newp[jm][1] : 0.f);
        if (polyak)
          img_store_pair(im.fwd + zoff + im.target_delta, im.fwd_chunks, r_even + jm, cb, newt[jm][0],
                         second ? newt[jm][1] : 0.f);
      }
    }
    if (im.bwd != nullptr && r_even < g.M) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = cb + e - im.bwd_col0;
        if (cb + e >= g.N || k < 0 || k >= im.bwd_cols) continue;
        const bool second = r_even + 1 < g.M;
        img_store_pair(im.bwd + zoff, im.bwd_chunks, k, r_even, newp[0][e], second ? newp[1][e] : 0.f);
        if (polyak)
          img_store_pair(im.bwd + zoff + im.target_delta, im.bwd_chunks, k, r_even, newt[0][e],
                         second ? newt[1][e] : 0.f);
      }
    }
  }
  stamp(5);
}

__global__ __launch_bounds__(64 * kTnWaves) void gemm_tn_group_kernel(GemmGroup G) {
  __shared__ float part[kTnWaves * 64 * kTnPart];
  kernarg_prefetch<sizeof(GemmGroup)>();
  int p = 0;
#pragma unroll
  for (int q = 1; q < kGemmGroupMax; ++q) p += (q < G.count && (int)blockIdx.x >= G.first[q]) ? 1 : 0;
  const GemmArgs& g = G.problem[p];
  const int tile = (int)blockIdx.x - G.first[p], tiles_n = (g.N + 31) / 32;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  // Tiles that reach past the last row / column of C still take the vector loads when the
  // operand ROWS are long enough (padded pitch: what is read beyond M / N only feeds outputs that
  // are never stored); clamped scalar loads only where a load would leave the row.
  const unsigned total = gridDim.x * gridDim.z;
  const bool sums = g.colsum != nullptr && tn == 0;               // uniform, like the next condition
  if (32 * tm + 32 > g.lda || 32 * tn + 32 > g.ldb) {
    if (sums) gemm_tn_tile<true, true>(g, tm, tn, part, G.adam, total, G.stamps);
    else gemm_tn_tile<true, false>(g, tm, tn, part, G.adam, total, G.stamps);
  } else {
    if (sums) gemm_tn_tile<false, true>(g, tm, tn, part, G.adam, total, G.stamps);
    else gemm_tn_tile<false, false>(g, tm, tn, part, G.adam, total, G.stamps);
  }
}

namespace {

// waves per tile: at most kGemmGroup chunks per wave, then more while the chip is not full
int waves_per_tile(int K, int64_t tiles_total) {
  const int chunks = (K + 15) / 16;
  int S = 1;
  while (S < kGemmMaxWaves && S * kGemmGroup < chunks) S *= 2;
  while (S < kGemmMaxWaves && S < chunks && tiles_total * S < 2048) S *= 2;
  return S;
}

}  // namespace

int launch_gemm_group(char mode_a, char mode_b, const GemmArgs* list, int count, int batch,
                      hipStream_t stream, const AdamFold* adam) {
  TONIC_REQUIRE(list && count >= 1 && count <= kGemmGroupMax && batch > 0,
                TONIC_ERR_INVALID_ARGUMENT, "gemm group: %d problems", count);
  GemmGroup G{};
  G.count = count;
  if (adam != nullptr) {
    G.adam = *adam;
    G.adam.on = 1;
  }
  int64_t tiles_total = 0;
  for (int p = 0; p < count; ++p) {
    const GemmArgs& g = list[p];
    TONIC_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K == list[0].K && g.K > 0,
                  TONIC_ERR_INVALID_ARGUMENT, "gemm group: problem %d (M=%d N=%d K=%d)", p, g.M,
                  g.N, g.K);
    G.problem[p] = g;
    tiles_total += (int64_t)((g.M + 15) / 16) * ((g.N + 31) / 32);
  }
  if (mode_a == 's' && mode_b == 's') {                // weight gradients: the interleaved TN kernel
    bool plain = true;
    for (int p = 0; p < count; ++p)
      plain = plain && !list[p].bias && !list[p].mask && list[p].act == ACT_NONE;
    if (plain) {
      int blocks = 0;
      for (int p = 0; p < count; ++p) {
        G.first[p] = blocks;
        blocks += ((list[p].M + 31) / 32) * ((list[p].N + 31) / 32);
      }
      G.first[count] = blocks;
      if (unsigned long long* base = g_forward_stamps.load()) {      // developer probe: ring of 8 launches
        static std::atomic<unsigned> launches{0};
        G.stamps = base + 8 * 16 + 8 * (launches.fetch_add(1) % 8);
      }
      hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(blocks, 1, batch), dim3(64, kTnWaves), 0,
                         stream, G);
      TONIC_CHECK_LAUNCH("gemm_tn_group");
      return TONIC_OK;
    }
  }
  TONIC_REQUIRE(adam == nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "gemm group: the optimizer epilogue needs the plain TN weight-gradient form");
  const int S = waves_per_tile(list[0].K, tiles_total * batch);
  const int per_block = S >= 4 ? 1 : 4 / S;
  int blocks = 0;
  for (int p = 0; p < count; ++p) {
    G.first[p] = blocks;
    const int tiles = ((list[p].M + 15) / 16) * ((list[p].N + 31) / 32);
    blocks += (tiles + per_block - 1) / per_block;
  }
  G.first[count] = blocks;
  const dim3 grid(blocks, 1, batch), block(64, S, per_block);
  const size_t lds = S > 1 ? (size_t)per_block * S * 64 * 9 * sizeof(float) : 0;
#define TONIC_GEMM_LAUNCH(AKC, BKC) \
  hipLaunchKernelGGL((gemm16_group_kernel<AKC, BKC>), grid, block, lds, stream, G)
  if (mode_a == 'c' && mode_b == 'c') TONIC_GEMM_LAUNCH(true, true);
  else if (mode_a == 'c' && mode_b == 's') TONIC_GEMM_LAUNCH(true, false);
  else if (mode_a == 's' && mode_b == 's') TONIC_GEMM_LAUNCH(false, false);
  else {
    set_error("gemm group: unsupported operand layouts '%c%c'", mode_a, mode_b);
    return TONIC_ERR_INVALID_ARGUMENT;
  }
#undef TONIC_GEMM_LAUNCH
  TONIC_CHECK_LAUNCH("gemm16_group");
  return TONIC_OK;
}

// ------------------------------------------------------------------------------ weight images (mlpimg.h)
// One workgroup = one (tile, k-chunk) block of one image, both terms: thread = (lane, pair of neighbouring k).
__global__ __launch_bounds__(256) void build_images_kernel(ImgJobs J) {
  int p = 0;
  for (int q = 1; q < J.count; ++q) p += (int)blockIdx.x >= J.first[q] ? 1 : 0;
  const ImgJob& j = J.job[p];
  const int blk = (int)blockIdx.x - J.first[p];
  const int c = blk % j.chunks, t = blk / j.chunks;
  const int pair = threadIdx.x & 3, lane = threadIdx.x >> 2;
  const int i = 16 * t + (lane & 15);                       // row of the A operand
  const int kk = 32 * c + 8 * (lane >> 4) + 2 * pair;       // its (even) k
  float w0 = 0.f, w1 = 0.f;
  if (!j.transposed) {
    if (i < j.rows) {
      const float* row = j.src + (int64_t)i * j.ld;
      if (kk < j.cols) w0 = row[kk];
      if (kk + 1 < j.cols) w1 = row[kk + 1];
    }
  } else if (i < j.ncols) {
    const float* col = j.src + j.col0 + i;
    if (kk < j.rows) w0 = col[(int64_t)kk * j.ld];
    if (kk + 1 < j.rows) w1 = col[(int64_t)(kk + 1) * j.ld];
  }
  unsigned hi, lo;
  img_terms(w0, w1, hi, lo);
  char* at = j.dst + ((int64_t)blk * 2 * 64 + lane) * 16 + 4 * pair;
  *reinterpret_cast<unsigned*>(at) = hi;
  *reinterpret_cast<unsigned*>(at + kImgTermBytes) = lo;
}

void ImgBuild::add(const float* src, int ld, int rows, int cols, char* block, ImgView v, bool transposed,
                   int col0, int ncols) {
  if (jobs.count >= kImgJobsMax) { jobs.count = kImgJobsMax + 1; return; }     // (launch_build_images refuses)
  ImgJob& j = jobs.job[jobs.count];
  j.src = src; j.ld = ld; j.rows = rows; j.cols = cols;
  j.transposed = transposed ? 1 : 0; j.col0 = col0; j.ncols = ncols;
  j.dst = block + v.off; j.tiles = v.tiles; j.chunks = v.chunks;
  jobs.first[jobs.count + 1] = jobs.first[jobs.count] + v.tiles * v.chunks;
  jobs.count += 1;
}

int launch_build_images(const ImgBuild& b, hipStream_t stream) {
  TONIC_REQUIRE(b.jobs.count >= 1 && b.jobs.count <= kImgJobsMax, TONIC_ERR_INVALID_ARGUMENT,
                "build_weight_images: %d conversions", b.jobs.count);
  hipLaunchKernelGGL(build_images_kernel, dim3(b.jobs.first[b.jobs.count]), dim3(256), 0, stream, b.jobs);
  TONIC_CHECK_LAUNCH("build_images_kernel");
  return TONIC_OK;
}

int launch_gemm(char mode_a, char mode_b, const GemmArgs& g, int batch, hipStream_t stream) {
  TONIC_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && batch > 0,
                TONIC_ERR_INVALID_ARGUMENT, "gemm: bad argument (M=%d N=%d K=%d)", g.M, g.N, g.K);
  if (mode_a == 's' && mode_b == 's' && !g.bias && !g.mask && g.act == ACT_NONE)
    return launch_gemm_group(mode_a, mode_b, &g, 1, batch, stream);
  const int tiles = ((g.M + 15) / 16) * ((g.N + 31) / 32);
  const int S = waves_per_tile(g.K, (int64_t)tiles * batch);
  const int per_block = S >= 4 ? 1 : 4 / S;         // tiles per workgroup (>= 4 waves each)
  const dim3 grid((tiles + per_block - 1) / per_block, 1, batch), block(64, S, per_block);
  const size_t lds = S > 1 ? (size_t)per_block * S * 64 * 9 * sizeof(float) : 0;
#define TONIC_GEMM_LAUNCH(AKC, BKC) \
  hipLaunchKernelGGL((gemm16_kernel<AKC, BKC>), grid, block, lds, stream, g)
  if (mode_a == 'c' && mode_b == 'c') TONIC_GEMM_LAUNCH(true, true);
  else if (mode_a == 'c' && mode_b == 's') TONIC_GEMM_LAUNCH(true, false);
  else if (mode_a == 's' && mode_b == 's') TONIC_GEMM_LAUNCH(false, false);
  else {
    set_error("gemm: unsupported operand layouts '%c%c'", mode_a, mode_b);
    return TONIC_ERR_INVALID_ARGUMENT;
  }
#undef TONIC_GEMM_LAUNCH
  TONIC_CHECK_LAUNCH("gemm16");
  return TONIC_OK;
}

}  // namespace tonic

using namespace tonic;

// Developer / test entry point: one GEMM of the building block (not used by the agents, which
// call the fused off-policy entry points).  mode = "NT", "NN" or "TN" as in gemm16.h.
extern "C" int tonic_gemm_f32(const char* mode, const float* d_a, const float* d_b, float* d_c,
                              const float* d_bias, const float* d_mask, float* d_colsum,
                              int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                              int32_t ldc, int32_t act, int32_t accumulate, double alpha,
                              void* stream) {
  TONIC_REQUIRE(mode != nullptr && mode[0] && mode[1], TONIC_ERR_INVALID_ARGUMENT,
                "tonic_gemm_f32: bad mode");
  GemmArgs g{};
  g.A = d_a; g.B = d_b; g.C = d_c; g.bias = d_bias; g.mask = d_mask; g.colsum = d_colsum;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = ldc;
  g.act = act; g.accumulate = accumulate; g.alpha = (float)alpha;
  const char a = mode[0] == 'N' ? 'c' : 's';
  const char b = mode[1] == 'T' ? 'c' : 's';
  return launch_gemm(a, b, g, 1, as_stream(stream));
}
