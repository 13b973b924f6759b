// Host <-> persistent-kernel signalling round trip over pinned host memory (gfx950):
// the host bumps a command word, one workgroup of a resident kernel polls it with system-scope
// loads, optionally reads `bytes` of pinned host data and writes `out_bytes` back, releases and
// answers through a second word.  Build: hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <vector>

__global__ void resident(volatile unsigned long long* cmd, unsigned* ack, const float* in,
                         float* out, int words_in, int words_out, int steps, int blocks_poll_host,
                         unsigned long long* relay, unsigned long long* clocks) {
  // shader clock while the kernel is resident: s_memtime ticks per 100 MHz wall-clock tick
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  __shared__ unsigned long long seen;
  __shared__ float sink[256];
  for (unsigned expect = 1; expect <= (unsigned)steps; ++expect) {
    if (threadIdx.x == 0) {
      unsigned long long c;
      if (blockIdx.x == 0 || blocks_poll_host) {
        do {
          c = __hip_atomic_load((unsigned long long*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } while ((unsigned)(c >> 32) != expect);
        if (blockIdx.x == 0 && !blocks_poll_host)
          __hip_atomic_store(relay, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        do {
          c = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(c >> 32) != expect);
      }
      seen = c;
    }
    __syncthreads();
    float acc = 0.f;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 30, 0x27000);
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < words_in; i += blockDim.x * gridDim.x)
      acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, i * 4, 0, 17));
    sink[threadIdx.x] = acc;
    __syncthreads();
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < words_out; i += blockDim.x * gridDim.x)
      out[i] = sink[(i + 1) & 255] + (float)expect;
    if (words_out > 0 && threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(ack + blockIdx.x, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = clock64() - c0; clocks[1] = wall_clock64() - w0; }
}

// optional load beside the resident kernel: does the power manager raise the shader clock?
__global__ void heater(float* out, const unsigned* stop) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
#pragma unroll
    for (int i = 0; i < 4096; ++i) a = __builtin_fmaf(a, b, 0.5f);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1, words_in = argc > 2 ? atoi(argv[2]) : 0,
            words_out = argc > 3 ? atoi(argv[3]) : 0, poll_host = argc > 4 ? atoi(argv[4]) : 0,
            heat_blocks = argc > 5 ? atoi(argv[5]) : 0;
  const int steps = 20000;
  char* host;
  hipHostMalloc((void**)&host, 1 << 20, hipHostMallocMapped);
  unsigned long long* cmd = (unsigned long long*)host;
  unsigned* ack = (unsigned*)(host + 256);
  float* in = (float*)(host + 4096);
  float* out = (float*)(host + 4096 + 262144);
  for (int i = 0; i < 65536; ++i) in[i] = (float)i;
  *cmd = 0;
  for (int b = 0; b < 64; ++b) ack[b] = 0;
  unsigned long long* relay;
  hipMalloc((void**)&relay, 64);
  hipMemset(relay, 0, 64);
  unsigned long long* clocks;
  hipMalloc((void**)&clocks, 16);
  unsigned* stop = (unsigned*)(host + 512);
  *stop = 0;
  hipStream_t side;
  hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  float* heat_out;
  hipMalloc((void**)&heat_out, 1024 * 256 * 4);
  if (heat_blocks > 0) hipLaunchKernelGGL(heater, dim3(heat_blocks), dim3(256), 0, side, heat_out, stop);
  hipLaunchKernelGGL(resident, dim3(blocks), dim3(256), 0, 0, cmd, ack, in, out, words_in,
                     words_out, steps, poll_host, relay, clocks);
  std::vector<double> lat(steps);
  for (unsigned s = 1; s <= (unsigned)steps; ++s) {
    const double t0 = now();
    __atomic_store_n(cmd, ((unsigned long long)s << 32) | 7u, __ATOMIC_RELEASE);
    for (int b = 0; b < blocks; ++b)
      while (__atomic_load_n(ack + b, __ATOMIC_ACQUIRE) != s) __builtin_ia32_pause();
    lat[s - 1] = now() - t0;
  }
  __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
  hipDeviceSynchronize();
  unsigned long long ck[2];
  hipMemcpy(ck, clocks, 16, hipMemcpyDeviceToHost);
  printf("shader clock while resident: %.0f MHz (heater blocks %d)\n", 100.0 * ck[0] / (double)ck[1], heat_blocks);
  std::sort(lat.begin() + 100, lat.end());
  printf("blocks %d in %d B out %d B poll_host %d: median %.2f us p10 %.2f p90 %.2f (out[1]=%g)\n",
         blocks, words_in * 4, words_out * 4, poll_host, lat[100 + (steps - 100) / 2] * 1e6,
         lat[100 + (steps - 100) / 10] * 1e6, lat[100 + 9 * (steps - 100) / 10] * 1e6, out[1]);
  return 0;
}
