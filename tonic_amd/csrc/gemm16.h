// Small-batch fp32 GEMM building block for the 256-wide off-policy networks (SAC / TD3).
//
// The learner batches are tiny (B = 100 .. 1024 rows, layers <= 256 wide): every GEMM is a few
// MFLOP and latency-bound, weights live in L2 / Infinity Cache.  So there is no LDS staging:
// one wave owns one 16 x 32 output tile (two v_mfma_f32_16x16x4_f32 accumulator chains — the
// instruction needs two independent chains per wave to issue back to back), and streams its
// operands straight from global memory into registers, 16 k-values per iteration.
//
// Operand access per lane (i = lane & 15, kg = lane >> 4), k-chunk c, element t = 0..3:
//   K-contiguous operand  P[row0 + i][16c + 4kg + t]   -> one 16-byte load (4 dwords if unaligned)
//   K-strided operand     P[16c + 4kg + t][col0 + i]   -> four dword loads, lanes i coalesce
// MFMA step t contracts k = 16c + 4kg + t from lane group kg for BOTH operands, so any mix of
// the two layouts is consistent (the k order of the chain is free).
//
//   NT  C[M,N] = A[M,K]  . B[N,K]^T   forward            (A k-contig, B k-contig)
//   NN  C[M,N] = A[M,K]  . B[K,N]     backward to input  (A k-contig, B k-strided)
//   TN  C[M,N] = A[K,M]^T. B[K,N]     weight gradient    (A k-strided, B k-strided)
//
// The weight gradients (TN without bias / mask / activation) have their own kernel
// (gemm_tn_group_kernel): both operands are contiguous along the OUTPUT index there, so a lane
// fetches two neighbouring output columns per load and a wave owns a 32 x 32 tile of 2 x 2
// interleaved MFMA tiles, four waves splitting K.
#pragma once
#include "common.h"

namespace tonic {

enum GemmAct : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;     // [N] added to every row (may be null)
  const float* mask;     // [M, ldmask]: C *= (mask > 0)  (ReLU derivative; may be null)
  float* colsum;         // TN only: colsum[m] (+)= sum_k A[k][m]  (bias gradient; may be null)
  int M, N, K;
  int lda, ldb, ldc, ldmask;
  int64_t strideA, strideB, strideC, strideBias, strideMask, strideColsum;   // batch (blockIdx.z)
  int act;               // GemmAct applied after bias
  int accumulate;        // C += result instead of C = result
  float alpha;           // result scale (applied before bias)
};

constexpr int kGemmGroupMax = 4;

struct GemmGroup {
  GemmArgs problem[kGemmGroupMax];
  int first[kGemmGroupMax + 1];      // workgroup ranges of the problems
  int count;
};

int launch_gemm(char mode_a, char mode_b, const GemmArgs& g, int batch, hipStream_t stream);
// `count` (<= kGemmGroupMax) GEMMs of the same layout and K in one launch (see gemm16.hip).
int launch_gemm_group(char mode_a, char mode_b, const GemmArgs* list, int count, int batch,
                      hipStream_t stream);

}  // namespace tonic
