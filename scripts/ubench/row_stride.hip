// L2 -> register streaming rate of the "MFMA A operand straight from global memory" pattern:
// lane (m = lane & 15, kg = lane >> 4) loads 16 bytes of row m at k-offset 16 * chunk + 4 * kg,
// four 16-row tiles per wave, four waves per workgroup, one workgroup per CU; the matrix (256
// rows) is shared by all workgroups and L2-resident.  Swept over the row stride.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void k(const float* W, int stride, int chunks, int reps, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4;
  const float* row[4];
  for (int j = 0; j < 4; ++j) row[j] = W + (size_t)(16 * (wave + 4 * j) + m) * stride + 4 * kg;
  f32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    const int shift = 4 * (r & 1);                  // the passes are not loop-invariant
    for (int c = 0; c < chunks; c += DEPTH) {
      f32x4 v[DEPTH][4];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[d][j] = *reinterpret_cast<const f32x4*>(row[j] + 16 * (c + d) + shift);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[d][j];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) sink[0] = acc[0];
}

template <int DEPTH>
void run(const float* W, float* sink, int stride, int blocks) {
  const int chunks = 16, reps = 64;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<DEPTH>, dim3(blocks), dim3(256), 0, 0, W, stride, chunks, reps, sink);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<DEPTH>, dim3(blocks), dim3(256), 0, 0, W, stride, chunks, reps, sink);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = 256.0 * chunks * 64 * reps;          // per workgroup per launch
  printf("stride %5d floats, %3d workgroups, %d chunks in flight: %6.1f B/ns per CU (%.1f us per 256 KB)\n",
         stride, blocks, DEPTH, bytes / (ms / 10 * 1e6), ms / 10 * 1e3 / reps);
}

int main() {
  float *W, *sink;
  (void)hipMalloc(&W, 256 * 400 * 4 + 4096); (void)hipMalloc(&sink, 64);
  (void)hipMemset(W, 0, 256 * 400 * 4 + 4096);
  for (int blocks : {64, 256}) {
    for (int stride : {256, 260, 272, 288, 320}) {
      run<2>(W, sink, stride, blocks);
      run<4>(W, sink, stride, blocks);
    }
  }
  return 0;
}
