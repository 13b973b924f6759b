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
      if (mask != nullptr && !(mask[(int64_t)m * g.ldmask + n] > 0.f)) v = 0.f;
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
                      hipStream_t stream) {
  TONIC_REQUIRE(list && count >= 1 && count <= kGemmGroupMax && batch > 0,
                TONIC_ERR_INVALID_ARGUMENT, "gemm group: %d problems", count);
  GemmGroup G{};
  G.count = count;
  int64_t tiles_total = 0;
  for (int p = 0; p < count; ++p) {
    const GemmArgs& g = list[p];
    TONIC_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K == list[0].K && g.K > 0,
                  TONIC_ERR_INVALID_ARGUMENT, "gemm group: problem %d (M=%d N=%d K=%d)", p, g.M,
                  g.N, g.K);
    G.problem[p] = g;
    tiles_total += (int64_t)((g.M + 15) / 16) * ((g.N + 31) / 32);
  }
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

int launch_gemm(char mode_a, char mode_b, const GemmArgs& g, int batch, hipStream_t stream) {
  TONIC_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && batch > 0,
                TONIC_ERR_INVALID_ARGUMENT, "gemm: bad argument (M=%d N=%d K=%d)", g.M, g.N, g.K);
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
