// Streaming rate of the W^T walk of the backward pass (operand element (feature i, k) lives at
// W[k][feature]): (B) one dword per lane and k, lanes i contiguous — 16 loads per 16-k chunk of
// four 16-feature tiles; (C) one 16-byte load per lane and k covering the four tiles when tile j
// owns features base + 4 i + j — 4 loads per chunk.  256 x 256 matrix, rows `stride` apart.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool WIDE>
__global__ __launch_bounds__(256) void k(const float* W, int stride, int reps, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    W += 4 * ((r & 1) ? 1 : -1) * (r > 0);          // the passes are not loop-invariant
    for (int c = 0; c < 16; c += 2) {
      if (WIDE) {
        f32x4 v[2][4];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[d][e] = *reinterpret_cast<const f32x4*>(W + (size_t)(16 * (c + d) + 4 * kg + e) * stride + 64 * wave + 4 * i);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += v[d][e];
      } else {
        float v[2][4][4];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[d][j][e] = W[(size_t)(16 * (c + d) + 4 * kg + e) * stride + 16 * (wave + 4 * j) + i];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[d][j][e];
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) sink[0] = acc[0];
}

template <bool WIDE>
void run(const float* W, float* sink, int stride, int blocks) {
  const int reps = 64;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<WIDE>, dim3(blocks), dim3(256), 0, 0, W, stride, reps, sink);
  (void)hipEventRecord(a, 0);
  for (int n = 0; n < 10; ++n) hipLaunchKernelGGL(k<WIDE>, dim3(blocks), dim3(256), 0, 0, W, stride, reps, sink);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%s stride %3d floats, %3d workgroups: %6.1f B/ns per CU (%.1f us per 256 KB)\n",
         WIDE ? "16-byte loads, interleaved tiles" : "dword loads                     ", stride, blocks,
         256.0 * 1024 * reps / (ms / 10 * 1e6), ms / 10 * 1e3 / reps);
}

int main() {
  float *W, *sink;
  (void)hipMalloc(&W, 256 * 400 * 4 + 4096); (void)hipMalloc(&sink, 64);
  (void)hipMemset(W, 0, 256 * 400 * 4 + 4096);
  for (int blocks : {64, 256})
    for (int stride : {256, 260}) { run<false>(W, sink, stride, blocks); run<true>(W, sink, stride, blocks); }
  return 0;
}
