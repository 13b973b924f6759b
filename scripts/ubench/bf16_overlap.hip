// Do VALU instructions hide behind bf16 MFMAs on gfx950 (they do not behind fp32 MFMAs)?
// K v_fma_f32 after every MFMA; cycles per MFMA slot; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MF, int K, int OP>
__global__ void k(float* out, int iters) {
  f32x4 acc[4] = {};
  float v[8];
  for (int t = 0; t < 8; ++t) v[t] = threadIdx.x * 1e-3f + t;
  const float b = 1.0f + blockIdx.x * 1e-5f;
  bf16x8 A8, B8; s16x4 A4, B4;
  for (int t = 0; t < 8; ++t) { A8[t] = (__bf16)(threadIdx.x * 1e-3f + t); B8[t] = (__bf16)(1.f + t); }
  for (int t = 0; t < 4; ++t) { A4[t] = threadIdx.x + t; B4[t] = 0x3f80 + t; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (MF == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A8, B8, acc[t], 0, 0, 0);
      if (MF == 1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A4, B4, acc[t], 0, 0, 0);
      if (MF == 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], b, acc[t], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const int r = q & 7;
        if (OP == 0) v[r] = __builtin_fmaf(v[r], b, 0.5f);
        if (OP == 1) v[r] = __builtin_amdgcn_exp2f(v[r]);
        if (OP == 2) { unsigned u = __float_as_uint(v[r]); u = (u & 0xffff0000u) + (unsigned)i; v[r] = __uint_as_float(u); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 8; ++t) s += v[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MF, int K, int OP>
void run(float* out, int iters, int threads, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, K, OP>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MF, K, OP>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const int wps = threads / 256;
  printf("  %-22s waves/SIMD=%d K=%d op=%d: %.1f cycles per MFMA per SIMD (2.4 GHz)\n", name, wps, K, OP,
         ms * 1e-3 * 2.4e9 / ((double)iters * 4 * wps));
}

template <int MF> void sweep(float* out, const char* name) {
  const int it = 20000;
  for (int threads : {256, 512}) {
    run<MF, 0, 0>(out, it, threads, name);
    run<MF, 1, 0>(out, it, threads, name);
    run<MF, 2, 0>(out, it, threads, name);
    run<MF, 3, 0>(out, it, threads, name);
    run<MF, 4, 0>(out, it, threads, name);
    run<MF, 6, 0>(out, it, threads, name);
    run<MF, 8, 0>(out, it, threads, name);
    run<MF, 1, 1>(out, it, threads, name);
    run<MF, 2, 1>(out, it, threads, name);
    run<MF, 4, 2>(out, it, threads, name);
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  sweep<0>(out, "16x16x32_bf16");
  sweep<1>(out, "16x16x16_bf16");
  sweep<2>(out, "16x16x4_f32");
  return 0;
}
