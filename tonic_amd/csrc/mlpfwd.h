// Fused forward of a two-hidden-layer ReLU MLP with up to two narrow heads: see mlpfwd.hip.
#pragma once
#include "gemm16.h"

namespace tonic {

struct MlpFwdArgs {
  const float* X;            // [B, ldx] inputs, K1 columns used
  int ldx, K1;
  const float* W1; const float* b1;   // [H, K1], [H]
  const float* W2; const float* b2;   // [H, H], [H]
  const float* Wh[2]; const float* bh[2];   // heads: [NH, H], [NH]
  int heads, NH;
  float* h1; float* h2;      // [B, H] hidden activations (written: the backward needs them)
  float* out[2];             // head outputs [B, ldo]
  int ldo;
  int act[2];                // GemmAct per head (ACT_NONE | ACT_TANH)
  int B, H;
  // batch of networks over blockIdx.y: element strides between consecutive networks
  int64_t stride_params, stride_hidden, stride_out;
};

bool mlp_forward_supported(int H, int NH, int heads);
int launch_mlp_forward(const MlpFwdArgs& a, int nets, hipStream_t stream);

}  // namespace tonic
