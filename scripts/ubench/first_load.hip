// How long until the first loads of a freshly launched kernel return?  A chain of dependent
// launches (like the per-step collect launches); every wave loads `n` 16-byte vectors per lane
// from an L2-resident buffer and stamps s_memrealtime (100 MHz) before and after.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(256) void k(const f32x4* src, float* sink, unsigned long long* stamps, int slot) {
  const unsigned long long t0 = wall_clock64();
  f32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = src[(i * 1024 + blockIdx.x * 256 + threadIdx.x) & 16383];
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += v[i][0] + v[i][3];
  if (s == 1.2345e-30f) sink[0] = s;
  const unsigned long long t1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = t1; }
}

template <int N>
void run(const f32x4* src, float* sink, unsigned long long* stamps, int blocks) {
  unsigned long long h[200];
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, src, sink, stamps, i);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
  double wait = 0, period = 0;
  for (int i = 50; i < 100; ++i) { wait += (h[2 * i + 1] - h[2 * i]) * 10.0; period += (h[2 * i] - h[2 * i - 2]) * 10.0; }
  printf("blocks %3d, %2d x 16 B per lane (%5.1f KB per block): loads return after %6.0f ns, launch period %6.0f ns\n",
         blocks, N, N * 256 * 16 / 1024.0, wait / 50, period / 50);
}

int main() {
  f32x4* src; float* sink; unsigned long long* stamps;
  (void)hipMalloc(&src, 16384 * 16); (void)hipMalloc(&sink, 64); (void)hipMalloc(&stamps, 1600);
  (void)hipMemset(src, 0, 16384 * 16);
  run<1>(src, sink, stamps, 16); run<4>(src, sink, stamps, 16); run<16>(src, sink, stamps, 16);
  run<32>(src, sink, stamps, 16); run<1>(src, sink, stamps, 1); run<16>(src, sink, stamps, 1);
  run<16>(src, sink, stamps, 64);
  return 0;
}
