// How fast does ONE wave run straight-line code it executes once every ~10 us (resident-kernel pattern)?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Unroll { template <class F> static __device__ __forceinline__ void go(F f) { f(); Unroll<N-1>::go(f);} };
template <> struct Unroll<0> { template <class F> static __device__ __forceinline__ void go(F) {} };

__global__ void k(unsigned long long* out, int steps, int idle_sleeps, float* sink, int dep) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float b = 1.0001f;
  for (int s = 0; s < steps; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    // 2048 straight-line fmas: 8 independent chains (dep = 0 semantics fixed at compile time)
#pragma unroll
    for (int i = 0; i < 2048; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0) out[s] = t1 - t0;
    for (int i = 0; i < idle_sleeps; ++i) __builtin_amdgcn_s_sleep(64);
  }
  float acc = 0; for (int i = 0; i < 8; ++i) acc += v[i];
  sink[threadIdx.x] = acc;
}
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8 * 1024); hipMalloc(&sink, 1024);
  for (int idle : {0, 10, 100, 400}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, 64, idle, sink, 0);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("idle sleeps %d: cycles for 2048 straight-line v_fma, steps 0,1,2,10,63: %llu %llu %llu %llu %llu\n", idle, h[0], h[1], h[2], h[10], h[63]);
  }
  return 0;
}
