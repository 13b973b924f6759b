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
#include <atomic>
#include "common.h"
#include "mlpimg.h"

namespace tonic {

enum GemmAct : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_ELU = 3 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;     // [N] added to every row (may be null)
  const float* mask;     // [M, ldmask]: C *= act'(z) given mask = act(z), by mask_act (may be null):
                         //   ACT_NONE / ACT_RELU: mask > 0;  ACT_TANH: 1 - mask^2;  ACT_ELU: mask > 0 ? 1 : mask + 1
  float* colsum;         // TN only: colsum[m] (+)= sum_k A[k][m]  (bias gradient; may be null)
  int M, N, K;
  int lda, ldb, ldc, ldmask;
  int64_t strideA, strideB, strideC, strideBias, strideMask, strideColsum;   // batch (blockIdx.z)
  int act;               // GemmAct applied after bias
  int mask_act;          // the activation whose derivative the mask stands for (see mask)
  int accumulate;        // C += result instead of C = result
  float alpha;           // result scale (applied before bias)
  ImgTarget img;         // TN + optimizer epilogue: the tensor's fp16x2 weight images (mlpimg.h; null pointers: none)
};

extern std::atomic<unsigned long long*> g_forward_stamps;    // developer probe, see tonic_debug_forward_stamps

constexpr int kGemmGroupMax = 4;

// Optional optimizer epilogue of the weight-gradient group (gemm_tn_group_kernel): the outputs of
// the group are the tensors of ONE network's flat gradient-sum block, so the offset of an output
// element from `grads` also addresses its parameter, its Adam moments and its target copy.  The
// workgroup that forms a tile of gradient sums applies torch.optim.Adam's step to the tile's
// parameters right away (same expressions as adam_kernel in optim.hip -> same bits) and, with
// `target`, the polyak update of the same entries; the last workgroup to arrive bumps the step
// counter and writes the logged statistics (adam_finalize).  Saves the optimizer launch and one
// round trip of the gradients through HBM.  Single rank, no gradient clipping (both need the
// complete gradient before the step).
struct AdamFold {
  const float* grads;        // base of the flat gradient-sum block the outputs live in
  float* params; float* exp_avg; float* exp_avg_sq;
  float* target;             // polyak target of the same block (null: none)
  int32_t* state;            // {step_count, stop_flag, -, arrivals}
  const float* consts;       // {step_size, bias_correction2_sqrt} of THIS step (null: formed from state[0])
  int64_t n;                 // parameters in the block: the 8 statistic sums follow at grads + n
  float grad_scale, beta2, eps, polyak_keep, polyak_mix;
  double beta1_d, beta2_d, lr_d;
  int stats_kind;            // 3 twin Q critics, 4 Q actor (see adam_finalize in optim.hip)
  float* info_row;
  // A chained launch ahead of this one may have given up on a value that never came (mlpfwd.h:
  // exchange_read): its failure word.  Non-zero = the gradient sums are not a gradient — no
  // parameter, moment or target is written, the step counter stays, the logged loss is NaN and
  // info_row[7] = 1 (agents.DDPG._update raises on it).  Null: always step.
  const unsigned* skip;
  int on;
};

struct GemmGroup {
  unsigned long long* stamps;        // developer probe: workgroup 0's phase stamps (null in the product path)
  GemmArgs problem[kGemmGroupMax];
  int first[kGemmGroupMax + 1];      // workgroup ranges of the problems
  int count;
  AdamFold adam;
};

int launch_gemm(char mode_a, char mode_b, const GemmArgs& g, int batch, hipStream_t stream);
// `count` (<= kGemmGroupMax) GEMMs of the same layout and K in one launch (see gemm16.hip).
int launch_gemm_group(char mode_a, char mode_b, const GemmArgs* list, int count, int batch,
                      hipStream_t stream, const AdamFold* adam = nullptr);

}  // namespace tonic
