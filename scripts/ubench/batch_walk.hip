// Streaming rate of the weight-gradient (TN) operand walk: a [1024, ld] batch-major matrix, a wave
// reads `W` consecutive floats per lane (16 lanes side by side) of rows 16 c + 4 kg + t — four
// rows, 4 rows apart, per instruction — all chunks c of its share (4 waves interleave them).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int W>
__global__ __launch_bounds__(256) void k(const float* P, int ld, int reps, float* sink) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kg = lane >> 4;
  const int col0 = (blockIdx.x % (256 / (16 * W))) * 16 * W + W * i;
  vec acc = 0;
  for (int r = 0; r < reps; ++r) {
    const float* Q = P + ((r & 1) ? W * 16 : 0);            // the passes are not loop-invariant
    for (int c = wave; c < 64; c += 16) {
      vec v[4][4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          v[d][t] = *reinterpret_cast<const vec*>(Q + (size_t)(16 * (c + 4 * d) + 4 * kg + t) * ld + col0);
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc += v[d][t];
    }
  }
  float s = 0;
  for (int e = 0; e < W; ++e) s += acc[e];
  if (s == 1.2345e-30f) sink[0] = s;
}

template <int W>
void run(const float* P, float* sink, int ld, int blocks) {
  const int reps = 32;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, P, ld, reps, sink);
  (void)hipEventRecord(a, 0);
  for (int n = 0; n < 10; ++n) hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, P, ld, reps, sink);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = 1024.0 * 16 * W * 4 * reps;            // per workgroup per launch
  printf("%2d B per lane, ld %3d, %3d workgroups: %6.1f B/ns per CU, %5.1f ns per load instruction per CU\n",
         4 * W, ld, blocks, bytes / (ms / 10 * 1e6), (ms / 10 * 1e6) / (reps * 64.0 * 4));
}

int main() {
  float *P, *sink;
  (void)hipMalloc(&P, 1024 * 400 * 4 + 4096); (void)hipMalloc(&sink, 64);
  (void)hipMemset(P, 0, 1024 * 400 * 4 + 4096);
  for (int ld : {256, 260, 264, 272, 288, 320}) {
    run<1>(P, sink, ld, 256); run<2>(P, sink, ld, 256); run<4>(P, sink, ld, 256);
  }
  return 0;
}
