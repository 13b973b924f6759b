// fp32 MFMA throughput against waves per SIMD and independent accumulators per wave (gfx950).
// Reports TFLOP/s over the whole chip (hipEvents over a long launch) for v_mfma_f32_16x16x4_f32
// and v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACCS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(float* out, int iters) {
  f32x4 acc[ACCS] = {};
  const float a = threadIdx.x * 1e-4f, b = 1.0f + blockIdx.x * 1e-5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int t = 0; t < ACCS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < ACCS; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int ACCS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, int iters) {
  f32x16 acc[ACCS] = {};
  const float a = threadIdx.x * 1e-4f, b = 1.0f + blockIdx.x * 1e-5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int t = 0; t < ACCS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < ACCS; ++t) s += acc[t][0] + acc[t][5];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <typename K>
void run(K kernel, const char* name, int accs, int waves, double flop_per_mfma, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(64 * waves), 0, 0, out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kernel, dim3(256), dim3(64 * waves), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 256.0 * waves * iters * 8 * accs;
  printf("%s  waves/SIMD %d  accumulators %d: %8.1f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", name,
         waves / 4, accs, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12,
         ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
  run(k16<1, 4>, "16x16x4", 1, 4, 2048, out);  run(k16<2, 4>, "16x16x4", 2, 4, 2048, out);
  run(k16<4, 4>, "16x16x4", 4, 4, 2048, out);  run(k16<8, 4>, "16x16x4", 8, 4, 2048, out);
  run(k16<4, 8>, "16x16x4", 4, 8, 2048, out);  run(k16<4, 16>, "16x16x4", 4, 16, 2048, out);
  run(k16<1, 16>, "16x16x4", 1, 16, 2048, out);
  run(k32<1, 4>, "32x32x2", 1, 4, 4096, out);  run(k32<2, 4>, "32x32x2", 2, 4, 4096, out);
  run(k32<2, 8>, "32x32x2", 2, 8, 4096, out);  run(k32<2, 16>, "32x32x2", 2, 16, 4096, out);
  return 0;
}
