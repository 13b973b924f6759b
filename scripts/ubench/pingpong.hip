// Host <-> persistent-kernel signalling round trip over pinned host memory (gfx950):
// the host bumps a command word, one workgroup of a resident kernel polls it with system-scope
// loads, optionally reads `bytes` of pinned host data and writes `out_bytes` back, releases and
// answers through a second word.  Build: hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong
// argv[6] = push: 0 the kernel PULLS command and input from pinned host memory; 1 / 2 the host PUSHES both into
// device memory through the BAR window (1: hipMalloc, 2: hipExtMallocWithFlags fine-grained) — write-combined
// stores, sfence, then the command word — and the kernel polls / reads its own HBM.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <vector>
#include <immintrin.h>
#include <string.h>
#include <sched.h>
#include <ctype.h>

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
    if (words_in > 1 && threadIdx.x == 0 && blockIdx.x == 0) {     // freshness of what the host wrote before the command
      const float first = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 17));
      const float last = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (words_in - 1) * 4, 0, 17));
      if (first != (float)expect || last != (float)expect) atomicAdd(&clocks[2], 1ull);
    }
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

// the process next to the GPU (what tonic_amd.parallel.bind_near_gpu does for the product): local_cpulist of the device
static void bind_near_gpu() {
  char bus[32] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) return;
  for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
  char path[128];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return;
  char list[512] = {0};
  if (!fgets(list, sizeof(list), f)) { fclose(f); return; }
  fclose(f);
  cpu_set_t set; CPU_ZERO(&set);
  for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a, b;
    if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; ++c) CPU_SET(c, &set); }
    else if (sscanf(tok, "%d", &a) == 1) CPU_SET(a, &set);
  }
  if (CPU_COUNT(&set) > 0) sched_setaffinity(0, sizeof(set), &set);
}

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1, words_in = argc > 2 ? atoi(argv[2]) : 0,
            words_out = argc > 3 ? atoi(argv[3]) : 0, poll_host = argc > 4 ? atoi(argv[4]) : 0,
            heat_blocks = argc > 5 ? atoi(argv[5]) : 0, push = argc > 6 ? atoi(argv[6]) : 0;
  const int steps = 20000;
  if (!getenv("PINGPONG_FAR")) bind_near_gpu();
  char* host;
  hipHostMalloc((void**)&host, 1 << 20, hipHostMallocMapped);
  unsigned long long* cmd = (unsigned long long*)host;
  unsigned* ack = (unsigned*)(host + 256);
  float* in = (float*)(host + 4096);
  float* out = (float*)(host + 4096 + 262144);
  for (int i = 0; i < 65536; ++i) in[i] = (float)i;
  *cmd = 0;
  float* staged = in;                       // push: what the host copies into the window every step
  if (push) {
    char* dev = nullptr;
    hipError_t e = push == 1 ? hipMalloc((void**)&dev, 1 << 20)
                   : hipExtMallocWithFlags((void**)&dev, 1 << 20, push == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    int large_bar = -1;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    printf("push %d: allocation %s, device pointer %p\n", push, hipGetErrorString(e), (void*)dev);
    hipMemset(dev, 0, 1 << 20);
    hipDeviceSynchronize();
    fflush(stdout);
    *(volatile unsigned*)(dev + 1024) = 0xabcd1234u;          // (a box without a CPU-visible window dies here)
    _mm_sfence();
    unsigned back = 0;
    hipMemcpy(&back, dev + 1024, 4, hipMemcpyDeviceToHost);
    printf("push %d: a host store into device memory reads back as %#x through hipMemcpy\n", push, back);
    fflush(stdout);
    cmd = (unsigned long long*)dev;
    in = (float*)(dev + 4096);
  }
  for (int b = 0; b < 64; ++b) ack[b] = 0;
  unsigned long long* relay;
  hipMalloc((void**)&relay, 64);
  hipMemset(relay, 0, 64);
  unsigned long long* clocks;
  hipMalloc((void**)&clocks, 32);
  hipMemset(clocks, 0, 32);
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
    if (push) {
      staged[0] = (float)s; staged[words_in > 0 ? words_in - 1 : 0] = (float)s;
      memcpy(in, staged, (size_t)words_in * 4);
      _mm_sfence();
      *(volatile unsigned long long*)cmd = ((unsigned long long)s << 32) | 7u;
      _mm_sfence();
    } else {
      in[0] = (float)s; in[words_in > 0 ? words_in - 1 : 0] = (float)s;
      __atomic_store_n(cmd, ((unsigned long long)s << 32) | 7u, __ATOMIC_RELEASE);
    }
    for (int b = 0; b < blocks; ++b)
      while (__atomic_load_n(ack + b, __ATOMIC_ACQUIRE) != s) __builtin_ia32_pause();
    lat[s - 1] = now() - t0;
  }
  __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
  hipDeviceSynchronize();
  unsigned long long ck[3];
  hipMemcpy(ck, clocks, 24, hipMemcpyDeviceToHost);
  printf("steps whose input was stale: %llu of %d\n", ck[2], steps);
  printf("shader clock while resident: %.0f MHz (heater blocks %d)\n", 100.0 * ck[0] / (double)ck[1], heat_blocks);
  std::sort(lat.begin() + 100, lat.end());
  printf("push %d blocks %d in %d B out %d B poll_host %d: median %.2f us p10 %.2f p90 %.2f (out[1]=%g)\n",
         push, blocks, words_in * 4, words_out * 4, poll_host, lat[100 + (steps - 100) / 2] * 1e6,
         lat[100 + (steps - 100) / 10] * 1e6, lat[100 + 9 * (steps - 100) / 10] * 1e6, out[1]);
  printf("  tail: p99 %.2f us p99.9 %.2f max %.2f\n", lat[100 + (size_t)(0.99 * (steps - 100))] * 1e6,
         lat[100 + (size_t)(0.999 * (steps - 100))] * 1e6, lat[steps - 1] * 1e6);
  return 0;
}
