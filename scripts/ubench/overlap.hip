// Micro-benchmark: how well do MFMA work and VALU work of TWO waves on one SIMD overlap on gfx950?
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_block(f32x4 (&acc)[4], float a, float b, int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
  }
}
__device__ __forceinline__ void valu_block(float (&v)[8], float m, int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(v[t], m, 0.5f);
  }
}

// mode 0: all MFMA   1: all VALU   2: waves 0-3 MFMA, 4-7 VALU
// mode 3: every wave alternates [mb MFMA groups][vb VALU groups], same phase
// mode 4: as 3, waves 4-7 start with the VALU phase (anti-phase)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mb, int vb) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {};
  float v[8];
  for (int t = 0; t < 8; ++t) v[t] = threadIdx.x * 1e-3f + t;
  const float a = threadIdx.x * 1e-4f, b = 1.0f + blockIdx.x * 1e-5f;
  if (MODE == 0) mfma_block(acc, a, b, iters * mb);
  if (MODE == 1) valu_block(v, b, iters * vb);
  if (MODE == 2) { if (wave < 4) mfma_block(acc, a, b, iters * mb); else valu_block(v, b, iters * vb); }
  if (MODE == 3 || MODE == 4) {
    if (MODE == 4 && wave >= 4) valu_block(v, b, vb);
    for (int i = 0; i < iters; ++i) { mfma_block(acc, a, b, mb); valu_block(v, b, vb); }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 8; ++t) s += v[t];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters, int mb, int vb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, mb, vb);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, mb, vb);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  // mb groups of 4 MFMA (4*32 = 128 pipe cycles each); vb groups of 8 v_fma (8*4 = 32 issue cycles each)
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int mb = 16, vb = cfg == 0 ? 32 : cfg == 1 ? 64 : 16;       // VALU issue = 0.5x, 1x, 0.25x of MFMA pipe time
    const double mf = 2.0 * iters * mb * 4 * 32, vf = 2.0 * iters * vb * 8 * 4;   // SIMD cycles, two waves
    printf("cfg mb=%d vb=%d: per-SIMD cycles needed: mfma %.0f, valu-issue %.0f\n", mb, vb, mf, vf);
    const float t0 = run<0>(out, iters, mb, vb), t1 = run<1>(out, iters, mb, vb), t2 = run<2>(out, iters, mb, vb);
    const float t3 = run<3>(out, iters, mb, vb), t4 = run<4>(out, iters, mb, vb);
    printf("  all-MFMA %.3f ms (%.2f GHz-equivalent)  all-VALU %.3f ms  split-roles %.3f ms\n", t0,
           mf / (t0 * 1e-3) / 1e9, t1, t2);
    printf("  alternating same-phase %.3f ms   anti-phase %.3f ms   (sum of parts %.3f, max %.3f)\n", t3, t4,
           t0 + t1, t0 > t1 ? t0 : t1);
  }
  return 0;
}
