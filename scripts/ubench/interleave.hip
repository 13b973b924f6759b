// Micro-benchmark: VALU instructions interleaved INTO the MFMA stream of one wave (K v_fma after every
// v_mfma_f32_16x16x4_f32), 1 or 2 waves per SIMD.  Does the VALU work hide in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k(float* out, int iters) {
  f32x4 acc[4] = {};
  float v[8];
  for (int t = 0; t < 8; ++t) v[t] = threadIdx.x * 1e-3f + t;
  const float a = threadIdx.x * 1e-4f, b = 1.0f + blockIdx.x * 1e-5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < K; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], b, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 8; ++t) s += v[t];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int K, int WAVES>
void run(float* out, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double mfma = (double)iters * 4 * (WAVES / 4);
  printf("  waves/SIMD %d  K=%2d VALU per MFMA: %.3f ms  -> %.1f cycles per MFMA slot at 2.4 GHz\n", WAVES / 4, K, ms,
         ms * 1e-3 * 2.4e9 / mfma);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  run<0, 4>(out, iters); run<2, 4>(out, iters); run<4, 4>(out, iters); run<6, 4>(out, iters);
  run<7, 4>(out, iters); run<8, 4>(out, iters); run<12, 4>(out, iters); run<16, 4>(out, iters);
  run<0, 8>(out, iters); run<4, 8>(out, iters); run<6, 8>(out, iters); run<8, 8>(out, iters); run<16, 8>(out, iters);
  return 0;
}
