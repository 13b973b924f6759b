// What bounds ONE wave walking a dependent float32 chain over 4096 rows whose operands come from LDS
// (the chain wave of gae_stream_kernel)?  Variants: registers only / + LDS reads and writes /
// + one barrier per 64 rows with idle partner waves; 7 or 4 dependent operations per row.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off chain.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OPS, bool LDS, bool BARRIER, int DIST, bool MASK = false>
__global__ __launch_bounds__(256) void k(float* out, int chunks, float lambda, float gamma) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, col = tid & 15;
  if (wave != 0) {
    if (BARRIER) for (int i = 0; i < chunks; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return;
  }
  float last = tid * 1e-3f;
  if (MASK && tid >= 16) {          // lanes 16 .. 63 only keep the barrier count (the wave executes s_barrier once)
    if (BARRIER) for (int i = 0; i < chunks; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return;
  }
  for (int i = 0; i < chunks; ++i) {
    const f32x4* quad = reinterpret_cast<const f32x4*>(lds + (i & 3) * 6144) + col;
    const f32x4* rew = reinterpret_cast<const f32x4*>(lds + (i & 3) * 6144 + 4096) + col;
    f32x4* ret_out = reinterpret_cast<f32x4*>(lds + 4 * 6144) + col;
    f32x4 q[3][4], r4[3];
    for (int b = 0; b < 3; ++b) { for (int p = 0; p < 4; ++p) q[b][p] = f32x4{0.1f, 1.f, 0.f, 1.f}; r4[b] = f32x4{0.5f, 0.25f, 0.125f, 0.3f}; }
    if (LDS) {
#pragma unroll
      for (int b = 0; b < DIST; ++b) {
#pragma unroll
        for (int p = 0; p < 4; ++p) q[b][p] = quad[(4 * b + p) * 16];
        r4[b] = rew[b * 16];
      }
    }
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) {
      const int cur = grp % 3, nxt = (grp + DIST) % 3;
      if (LDS && grp + DIST < 16) {
#pragma unroll
        for (int p = 0; p < 4; ++p) q[nxt][p] = quad[(4 * (grp + DIST) + p) * 16];
        r4[nxt] = rew[(grp + DIST) * 16];
      }
      f32x4 ret;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float boot = q[cur][p][0] + lambda * last;
        if (OPS == 7) { boot = boot * q[cur][p][1]; boot = boot + q[cur][p][2]; boot = boot * q[cur][p][3]; }
        last = r4[cur][p] + gamma * boot;
        ret[p] = last;
      }
      if (LDS) ret_out[grp * 16] = ret;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  out[blockIdx.x * 64 + tid] = last;
}

template <int OPS, bool LDS, bool BARRIER, int DIST, bool MASK = false>
void run(const char* label, float* out) {
  const size_t lds = (4 * 6144 + 1024 * 4) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<OPS, LDS, BARRIER, DIST, MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<OPS, LDS, BARRIER, DIST, MASK>), dim3(16), dim3(256), lds, 0, out, 64, 0.97f, 0.99f);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s %7.1f us per 4096 rows  (%.1f ns per row)\n", label, ms * 100.f, ms * 100.f * 1000.f / 4096.f);
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
  run<7, false, false, 2>("7 ops, registers only", out);
  run<4, false, false, 2>("4 ops, registers only", out);
  run<7, true, false, 1>("7 ops, LDS operands 1 group ahead", out);
  run<7, true, false, 2>("7 ops, LDS operands 2 groups ahead", out);
  run<4, true, false, 2>("4 ops, LDS operands 2 groups ahead", out);
  run<7, true, true, 2>("7 ops, LDS 2 ahead, barrier per 64 rows", out);
  run<4, true, true, 2>("4 ops, LDS 2 ahead, barrier per 64 rows", out);
  run<7, true, false, 2, true>("7 ops, LDS 2 ahead, lanes 16..63 masked off", out);
  run<4, true, false, 2, true>("4 ops, LDS 2 ahead, lanes 16..63 masked off", out);
  return 0;
}
