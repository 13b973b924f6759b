// Declarations shared by the two fused-MLP kernel families (mlp64.hip: 32x32x2 tiles, one
// wave per SIMD; mlp64x16.hip: 16x16x4 tiles, two waves per SIMD).
#pragma once
#include "common.h"

namespace tonic {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;   // log(sqrt(2*pi))
constexpr float kEntropyConst = 1.41893853320467274178f; // 0.5 + 0.5*log(2*pi)

struct MlpArgs {
  const float* params;
  const float* obs;
  const float* actions;     // actor grad / unused
  const float* adv;         // raw advantages
  const float* adv_stats;   // {mean, std, all_zero, normalise}
  const float* old_logp;
  const float* returns;     // critic grad
  const float* norm_mean;   // critic
  const float* norm_std;
  float norm_clip;          // MeanStd(clip=...): clamp bound of the normalised input (+inf: none)
  const float* eps;         // act
  float* out0;              // act: actions, value: values, grad: partials
  float* out1;              // act: log_probs
  const int32_t* skip;
  int64_t n;
  int O, A;
  float clip_lo, clip_hi;
  int plain;                // 1: plain policy gradient, loss = -mean(adv * logp) (actors.py:20-51)
  int pstride;
  int skew;                 // initial phase offset (x ~8k cycles) of the second wave per SIMD
  int prio;                 // wave priorities of the 16-sample grad kernels (tuning key "grad_prio")
};

// tanh(x) = sign(x) (1 - t) / (1 + t), t = exp(-2|x|): one v_exp_f32 + one v_rcp_f32, no
// branches (the libm tanhf is ~40 instructions with a divergent branch).  Absolute error
// <= ~2e-7 over the whole range, i.e. float32 rounding level of the surrounding dot products.
__device__ __forceinline__ float tanh_fast(float x) {
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.8853900817779268f);   // 2*log2(e)
  const float y = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
  return copysignf(y, x);
}

// mlp64.hip: fixed-order fold of the per-workgroup partial images into d_grad_sums (+ the
// log_scale / entropy / sigma post-processing of the actor)
// (log_scale_offset: where the actor's log_scale sits in the block; < 0: the default torso's place)
int launch_reduce_partials(bool actor, const float* partials, int blocks, int pstride, int P,
                           const float* params, float* d_grad_sums, int O, int A,
                           float entropy_coeff, double rows, const int32_t* skip,
                           hipStream_t stream, int log_scale_offset = -1);

// mlpwide.hip: the layer-by-layer path — shapes outside the fused kernels (O > 32 or A > 8) and ANY torso
// MLP(sizes, activation) of tonic/torch/models/utils.py:4-23 (1 .. 4 hidden layers of 4 .. 384 units,
// multiples of 4, Tanh or ReLU); the default argument is the reference's default torso (a2c.py:7-17).
constexpr int kMaxTorsoLayers = 4;
struct Torso {
  int layers;
  int size[kMaxTorsoLayers];
  int act;                                   // 1 Tanh, 2 ReLU
  static Torso standard() { return Torso{2, {64, 64, 0, 0}, 1}; }
};
bool torso_supported(const Torso& t);
int64_t torso_param_count(int O, int A, bool actor, const Torso& t);
bool wide_shape(int O, int A, bool actor);
bool wide_supported(int O, int A, bool actor);
int64_t wide_workspace_bytes(int64_t n, int O, int A, bool actor, const Torso& t = Torso::standard());
int wide_actor_grad(const MlpArgs& a, float* d_grad_sums, float entropy_coeff, void* d_workspace,
                    int64_t workspace_bytes, hipStream_t stream, const Torso& t = Torso::standard());
int wide_critic_grad(const MlpArgs& a, float* d_grad_sums, void* d_workspace,
                     int64_t workspace_bytes, hipStream_t stream, const Torso& t = Torso::standard());
int wide_act(const MlpArgs& a, void* d_workspace, int64_t workspace_bytes, hipStream_t stream,
             const Torso& t = Torso::standard());
int wide_value(const MlpArgs& a, void* d_workspace, int64_t workspace_bytes, hipStream_t stream,
               const Torso& t = Torso::standard());

// mlp64x16.hip
bool grad16_supported(int O, int A, bool actor);
int grad16_blocks(int64_t n);
int launch_grad16(bool actor, int blocks, hipStream_t stream, const MlpArgs& args, int chain);
int launch_grad16_probe(int blocks, hipStream_t stream, const MlpArgs& args);
int launch_values16(int blocks, hipStream_t stream, const MlpArgs& args, int chain);

}  // namespace tonic
