// Can the host store into hipMalloc'ed memory directly (large BAR)?  And what does a command word
// in device memory buy over one in pinned host memory for a resident kernel's poll?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <algorithm>
#include <vector>
__global__ void resident(volatile unsigned long long* cmd, unsigned* ack, int steps) {
  for (unsigned expect = 1; expect <= (unsigned)steps; ++expect) {
    unsigned long long c;
    do {
      c = __hip_atomic_load((unsigned long long*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } while ((unsigned)(c >> 32) != expect);
    __hip_atomic_store(ack, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: cmd in pinned host memory, 1: in device memory
  const int steps = 20000;
  char* host; hipHostMalloc((void**)&host, 4096, hipHostMallocMapped);
  unsigned* ack = (unsigned*)(host + 256); *ack = 0;
  unsigned long long* cmd;
  if (mode == 0) { cmd = (unsigned long long*)host; *cmd = 0; }
  else {
    hipError_t e = hipExtMallocWithFlags((void**)&cmd, 4096, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocDefault);
    printf("alloc mode %d: %s, ptr %p\n", mode, hipGetErrorString(e), (void*)cmd);
    hipMemset(cmd, 0, 4096); hipDeviceSynchronize();
    hipPointerAttribute_t a; hipPointerGetAttributes(&a, cmd);
    printf("type %d host ptr %p dev ptr %p\n", (int)a.type, a.hostPointer, a.devicePointer);
    fflush(stdout);
  }
  hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, 0, cmd, ack, steps);
  std::vector<double> lat(steps);
  for (unsigned s = 1; s <= (unsigned)steps; ++s) {
    const double t0 = now();
    __atomic_store_n(cmd, ((unsigned long long)s << 32) | 7u, __ATOMIC_RELEASE);
    while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != s) __builtin_ia32_pause();
    lat[s - 1] = now() - t0;
  }
  hipDeviceSynchronize();
  std::sort(lat.begin() + 100, lat.end());
  printf("mode %d: median %.2f us p10 %.2f p90 %.2f\n", mode, lat[100 + (steps - 100) / 2] * 1e6,
         lat[100 + (steps - 100) / 10] * 1e6, lat[100 + 9 * (steps - 100) / 10] * 1e6);
  return 0;
}
