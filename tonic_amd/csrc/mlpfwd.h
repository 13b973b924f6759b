// Fused forward of a two-hidden-layer ReLU MLP with up to two narrow heads: see mlpfwd.hip.
#pragma once
#include "gemm16.h"

namespace tonic {

// Parameter layout of the off-policy networks (see include/tonic_hip.h, "off-policy parameter
// layout"): tensors in parameters() order, every tensor starts on a 16-byte boundary, and the
// rows of a [rows, cols] weight are weight_ld(cols) floats apart — a multiple of 4 floats that
// is NOT a multiple of 32.  The kernels stream MFMA operands straight from L2, lane = row: with
// rows a multiple of 128 bytes apart all 16 rows of a load hit the same L1 bank group and the
// stream runs at 38 B/ns per CU instead of 70 - 84 (profiles/r01_ubench_row_stride.md), and
// rows that are not 16-byte aligned split every 16-byte load.  The padding floats are zero and
// stay zero (zero gradient -> Adam moves nothing; polyak of zeros).
__host__ __device__ inline int weight_ld(int cols) {
  const int ld = (cols + 3) / 4 * 4;
  return ld % 32 == 0 ? ld + 4 : ld;
}
__host__ __device__ inline int64_t slot4(int64_t floats) { return (floats + 3) / 4 * 4; }

struct MlpFwdArgs {
  const float* X;            // [B, ldx] inputs, K1 columns used
  int ldx, K1;
  const float* W1; const float* b1;   // [H, K1] rows ldw1 apart, [H]
  const float* W2; const float* b2;   // [H, H] rows ldw2 apart, [H]
  const float* Wh[2]; const float* bh[2];   // heads: [NH, H] rows ldw2 apart, [NH]
  int ldw1, ldw2;            // weight row strides (weight_ld): multiples of 4, 16-byte aligned rows
  int heads, NH;
  float* h1; float* h2;      // [B, H] hidden activations, rows ldh apart (written: the backward needs them)
  int ldh;                   // >= H, multiple of 4 (weight_ld(H): the weight-gradient GEMMs walk these rows)
  float* out[2];             // head outputs [B, ldo]
  int ldo;
  int act[2];                // GemmAct per head (ACT_NONE | ACT_TANH)
  int B, H;
  // batch of networks over blockIdx.y: element strides between consecutive networks
  int64_t stride_params, stride_hidden, stride_out;
  // networks >= split read a second parameter set and a second input (target critics on
  // (s', a') and online critics on (s, a) in one launch): parameter pointers move by
  // second_params ELEMENTS on top of the linear stride, the input is X2.  split >= nets: unused.
  int split;
  int64_t second_params;
  const float* X2;
};

// Input-gradient chain of the same network (see mlp_backward_kernel in mlpfwd.hip).
struct MlpBwdArgs {
  int heads;                 // 0: critic (dq, w3); 1..2: actor heads (dhead, Wh)
  const float* dq;           // [B] per network            (heads == 0)
  const float* w3;           // [H]                        (heads == 0)
  const float* dhead[2];     // [B, ldh] gradients at the head outputs
  const float* Wh[2];        // [NH, H] rows ldw2 apart
  int NH, ldh;
  const float* W2;           // [H, H] rows ldw2 apart
  const float* W1;           // [H, K1] rows ldw1 apart (only for dxa)
  int ldw1, ldw2;
  int K1, xa_first, xa_count;   // dxa = columns [xa_first, xa_first + xa_count) of dz1 . W1
  const float* h1; const float* h2;   // [B, H] forward activations (ReLU masks), rows ldhid apart
  float* dz2; float* dz1;    // [B, H] outputs, rows ldhid apart
  int ldhid;
  float* dxa;                // [B, ldxa]
  int ldxa;
  int B, H;
  int64_t stride_params, stride_hidden, stride_dq, stride_dxa;
};

bool mlp_forward_supported(int H, int NH, int heads);
bool mlp_backward_supported(int H, int NH, int heads, int xa_count);
int launch_mlp_backward(const MlpBwdArgs& a, int nets, hipStream_t stream);
int launch_mlp_forward(const MlpFwdArgs& a, int nets, hipStream_t stream);

}  // namespace tonic
