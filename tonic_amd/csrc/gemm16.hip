// gemm16: see gemm16.h.  gfx950, v_mfma_f32_16x16x4_f32, one wave per 16 x 32 output tile.
#include "gemm16.h"

namespace tonic {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four k-values (k = k0 + 4*kg + t) of one operand row/column `idx` for this lane.
template <bool KC>
__device__ __forceinline__ void load_operand(const float* __restrict__ P, int ld, int idx,
                                             int idx_limit, int kbase, int K, bool vec_ok,
                                             float (&v)[4]) {
  const bool in = idx < idx_limit;
  if (KC) {
    const float* p = P + (int64_t)idx * ld + kbase;
    if (in && vec_ok && kbase + 4 <= K) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p);
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = (in && kbase + t < K) ? p[t] : 0.f;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      v[t] = (in && kbase + t < K) ? P[(int64_t)(kbase + t) * ld + idx] : 0.f;
  }
}

// SPLITK = false: each of the 4 waves of a workgroup owns its own 16 x 32 output tile.
// SPLITK = true : the 4 waves share ONE tile and interleave the k-chunks (long contractions such
//                 as the weight gradients, K = batch size); partial tiles are combined through
//                 LDS in wave order (deterministic).
template <bool A_KC, bool B_KC, bool SPLITK>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int tiles_n = (g.N + 31) / 32, tiles_m = (g.M + 15) / 16;
  const int tile = SPLITK ? blockIdx.x : blockIdx.x * 4 + wave;
  if (tile >= tiles_m * tiles_n) return;          // uniform per workgroup when SPLITK
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 16, n0 = tn * 32;
  const int z = blockIdx.z;
  const float* A = g.A + z * g.strideA;
  const float* B = g.B + z * g.strideB;
  float* C = g.C + z * g.strideC;
  const bool vec_a = A_KC && (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool vec_b = B_KC && (g.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float colsum = 0.f;
  const int kstep = SPLITK ? 64 : 16;
  int k0 = SPLITK ? 16 * wave : 0;
  // register double buffering: chunk c+1 is in flight while chunk c feeds the MFMAs
  float a[4], b0[4], b1[4], an[4], b0n[4], b1n[4];
  if (k0 < g.K) {
    load_operand<A_KC>(A, g.lda, m0 + i, g.M, k0 + 4 * kg, g.K, vec_a, a);
    load_operand<B_KC>(B, g.ldb, n0 + i, g.N, k0 + 4 * kg, g.K, vec_b, b0);
    load_operand<B_KC>(B, g.ldb, n0 + 16 + i, g.N, k0 + 4 * kg, g.K, vec_b, b1);
  }
  for (; k0 < g.K; k0 += kstep) {
    const int kn = k0 + kstep + 4 * kg;
    const bool more = k0 + kstep < g.K;
    if (more) {
      load_operand<A_KC>(A, g.lda, m0 + i, g.M, kn, g.K, vec_a, an);
      load_operand<B_KC>(B, g.ldb, n0 + i, g.N, kn, g.K, vec_b, b0n);
      load_operand<B_KC>(B, g.ldb, n0 + 16 + i, g.N, kn, g.K, vec_b, b1n);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = mfma16(a[t], b0[t], acc0);
      acc1 = mfma16(a[t], b1[t], acc1);
    }
    if (!A_KC) colsum += (a[0] + a[1]) + (a[2] + a[3]);
    if (more) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = an[t]; b0[t] = b0n[t]; b1[t] = b1n[t]; }
    }
  }

  if (SPLITK) {
    __shared__ float part[4][64][9];
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[wave][lane][r] = acc0[r]; part[wave][lane][4 + r] = acc1[r]; }
    part[wave][lane][8] = colsum;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc0[r] = (part[0][lane][r] + part[1][lane][r]) + (part[2][lane][r] + part[3][lane][r]);
      acc1[r] = (part[0][lane][4 + r] + part[1][lane][4 + r]) +
                (part[2][lane][4 + r] + part[3][lane][4 + r]);
    }
    colsum = (part[0][lane][8] + part[1][lane][8]) + (part[2][lane][8] + part[3][lane][8]);
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

int launch_gemm(char mode_a, char mode_b, const GemmArgs& g, int batch, hipStream_t stream) {
  TONIC_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && batch > 0,
                TONIC_ERR_INVALID_ARGUMENT, "gemm: bad argument (M=%d N=%d K=%d)", g.M, g.N, g.K);
  const int tiles = ((g.M + 15) / 16) * ((g.N + 31) / 32);
  const bool splitk = g.K >= 256 && tiles <= 1024;      // long contraction, few tiles
  const dim3 grid(splitk ? tiles : (tiles + 3) / 4, 1, batch), block(256);
#define TONIC_GEMM_LAUNCH(AKC, BKC)                                                          \
  do {                                                                                       \
    if (splitk) hipLaunchKernelGGL((gemm16_kernel<AKC, BKC, true>), grid, block, 0, stream, g);  \
    else hipLaunchKernelGGL((gemm16_kernel<AKC, BKC, false>), grid, block, 0, stream, g);       \
  } while (0)
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
