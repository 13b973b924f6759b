// Which VALU instruction classes steal cycles from a stream of fp32 MFMAs?  K instructions of one class are
// placed after every v_mfma_f32_16x16x4_f32 (one wave per SIMD); cycles per MFMA slot are reported.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP, int K>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f32x4 acc[4] = {};
  float v[8]; unsigned u[8]; f32x2 p[8];
  for (int t = 0; t < 8; ++t) { v[t] = threadIdx.x * 1e-3f + t; u[t] = threadIdx.x + t; p[t] = f32x2{v[t], v[t] + 1.f}; }
  const float a = threadIdx.x * 1e-4f, b = 1.0f + blockIdx.x * 1e-5f;
  const f32x2 b2 = {b, b}, h2 = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const int r = q & 7;
        if (OP == 0) v[r] = __builtin_fmaf(v[r], b, 0.5f);
        if (OP == 1) u[r] = u[r] * 3u + 7u;                      // v_mad_u32_u24 / mul_lo + add
        if (OP == 2) u[r] = (u[r] ^ 0x5a5a5a5au) + (unsigned)i;  // xor + add (two int ops)
        if (OP == 3) v[r] = __builtin_amdgcn_exp2f(v[r]);        // transcendental
        if (OP == 4) p[r] = __builtin_elementwise_fma(p[r], b2, h2);   // v_pk_fma_f32
        if (OP == 5) v[r] = __builtin_amdgcn_rcpf(v[r]);
        if (OP == 6) v[r] = v[r] + b;                            // v_add_f32
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 8; ++t) s += v[t] + (float)u[t] + p[t][0] + p[t][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int K>
void run(float* out, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, K>), dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP, K>), dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("  %-28s K=%d: %.1f cycles per MFMA slot (2.4 GHz)\n", name, K, ms * 1e-3 * 2.4e9 / ((double)iters * 4));
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int it = 20000;
  run<0, 0>(out, it, "baseline (MFMA only)");
  run<0, 4>(out, it, "v_fma_f32"); run<0, 8>(out, it, "v_fma_f32");
  run<6, 4>(out, it, "v_add_f32"); run<6, 8>(out, it, "v_add_f32");
  run<1, 4>(out, it, "int mul+add"); run<1, 8>(out, it, "int mul+add");
  run<2, 4>(out, it, "int xor+add (2 ops)"); run<2, 8>(out, it, "int xor+add (2 ops)");
  run<3, 2>(out, it, "v_exp_f32"); run<3, 4>(out, it, "v_exp_f32");
  run<5, 2>(out, it, "v_rcp_f32"); run<5, 4>(out, it, "v_rcp_f32");
  run<4, 4>(out, it, "v_pk_fma_f32"); run<4, 8>(out, it, "v_pk_fma_f32");
  return 0;
}
