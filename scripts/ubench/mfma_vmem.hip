// Micro-benchmark: does the weight stream of the fused off-policy forward overlap its MFMAs?
// One workgroup of four waves per CU (256 workgroups), every wave plays one 256 x 256 layer of
// mlp_forward_kernel per repetition: 16 k-chunks x 4 tiles = 64 loads of 16 bytes per lane (64 KB
// per wave, 256 KB per workgroup, L2-resident) and 256 v_mfma_f32_16x16x4_f32.
//   mode 0  MFMAs only (operands in registers)
//   mode 1  loads only (kept alive with an empty asm, no VALU work)
//   mode 2  both, software-pipelined like Layer::run (the loads of the next two chunks are issued,
//           then the MFMAs of the current two run)
// pattern 0: rows 1040 bytes apart, lane (m, kg) reads 16 bytes of row m at 16 c + 4 kg  (today)
// pattern 1: rows 1024 bytes apart (the dense layout)
// pattern 2: operand-order image: chunk c of tile j is ONE contiguous 1 KB block, lane l reads
//            bytes [16 l, 16 l + 16)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k(const float* W, int reps, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
  const int stride = PATTERN == 0 ? 260 : 256;
  const float* base[4];
  for (int j = 0; j < 4; ++j) {
    const int tile = wave + 4 * j;
    base[j] = PATTERN == 2 ? W + (size_t)tile * 16 * 256 + 4 * lane
                           : W + (size_t)(16 * tile + m) * stride + 4 * kg;
  }
  const int step = PATTERN == 2 ? 256 : 16;                 // floats between consecutive chunks
  f32x4 acc[4] = {};
  const float bop = 1.0f + lane * 1e-3f;
  f32x4 wa[2][4], wb[2][4];
  auto fill = [&](f32x4 (&w)[2][4], int c) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        w[q][j] = MODE == 0 ? f32x4{bop, bop, bop, bop}
                            : *reinterpret_cast<const f32x4*>(base[j] + (size_t)((c + q) & 15) * step);
  };
  auto compute = [&](const f32x4 (&w)[2][4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(w[q][j]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q][j][e], bop, acc[j], 0, 0, 0);
      }
    }
  };
  // mode 3: like 2, but the eight loads of the next set are issued one by one, each after four
  // MFMAs of the current set
  auto weave = [&](const f32x4 (&w)[2][4], f32x4 (&next)[2][4], int c) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q][j][e], bop, acc[j], 0, 0, 0);
        const int i = 4 * q + e;
        next[i >> 2][i & 3] =
            *reinterpret_cast<const f32x4*>(base[i & 3] + (size_t)((c + (i >> 2)) & 15) * step);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  if (MODE == 3) {
    for (int r = 0; r < reps; ++r) {
      fill(wa, 0);
      for (int c = 0; c < 16; c += 4) {
        weave(wa, wb, c + 2);
        weave(wb, wa, c + 4);
      }
    }
  } else
  for (int r = 0; r < reps; ++r) {
    fill(wa, 0);
    for (int c = 0; c < 16; c += 4) {
      fill(wb, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(wa);
      __builtin_amdgcn_sched_barrier(0);
      fill(wa, c + 4);
      __builtin_amdgcn_sched_barrier(0);
      compute(wb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 1.2345e-30f) sink[0] = s;
}

template <int MODE, int PATTERN>
void run(const float* W, float* sink) {
  const int reps = 200;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(256), dim3(256), 0, 0, W, reps, sink);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(256), dim3(256), 0, 0, W, reps, sink);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const char* mode[4] = {"MFMA only ", "loads only", "both      ", "both woven"};
  const char* pattern[3] = {"rows 1040 B apart", "rows 1024 B apart", "operand-order 1 KB blocks"};
  printf("%s  %-26s %6.2f us per 256 x 256 layer and workgroup\n", mode[MODE], MODE == 0 ? "-" : pattern[PATTERN],
         ms / 5 * 1e3 / reps);
}

int main() {
  float *W, *sink;
  (void)hipMalloc(&W, 300 * 400 * 4); (void)hipMalloc(&sink, 64);
  (void)hipMemset(W, 0, 300 * 400 * 4);
  run<0, 0>(W, sink);
  run<1, 0>(W, sink); run<2, 0>(W, sink); run<3, 0>(W, sink);
  run<1, 1>(W, sink); run<2, 1>(W, sink); run<3, 1>(W, sink);
  run<1, 2>(W, sink); run<2, 2>(W, sink); run<3, 2>(W, sink);
  return 0;
}
