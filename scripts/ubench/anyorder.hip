// Does hipExtAnyOrderLaunch let a kernel start beside the one in front of it in the SAME stream (no AQL barrier
// bit) on gfx950?  Two one-workgroup kernels that spin for `us` microseconds each, back to back:
//   serial ~ 2 us, overlapped ~ 1 us.   build: hipcc -O2 --offload-arch=gfx950 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>

__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out) *out = wall_clock64();
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned long long* d;
  hipMalloc(&d, 64);
  const unsigned long long ticks = 5000;      // 50 us of the 100 MHz clock
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 20; ++k) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, d);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, ticks, d + 1);
        else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr,
                                   mode == 1 ? hipExtAnyOrderLaunch : 0, ticks, d + 1);
      }
      hipStreamSynchronize(st);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%s: %.1f us per pair of 50 us kernels\n",
             mode == 0 ? "hipLaunchKernelGGL x2      " : mode == 1 ? "second: hipExtAnyOrderLaunch" : "second: hipExtLaunch flags 0", us / 20);
    }
  }
  return 0;
}
