// Shared host/device helpers for libtonic_hip.so (gfx950 only).
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tonic_hip.h"
#include "../../include/tonic_hip_dev.h"

namespace tonic {

void set_error(const char* fmt, ...);

#define TONIC_REQUIRE(cond, code, ...)      \
  do {                                      \
    if (!(cond)) {                          \
      ::tonic::set_error(__VA_ARGS__);      \
      return (code);                        \
    }                                       \
  } while (0)

#define TONIC_CHECK_LAUNCH(what)                                               \
  do {                                                                         \
    hipError_t err__ = hipGetLastError();                                      \
    if (err__ != hipSuccess) {                                                 \
      ::tonic::set_error("%s: %s", (what), hipGetErrorString(err__));          \
      return TONIC_ERR_LAUNCH;                                                 \
    }                                                                          \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kHidden = 64;       // both hidden layers of the PPO actor/critic (a2c.py:7-17)
constexpr int kStatSlots = 8;     // statistic sums appended to every gradient block

// Orders LDS traffic of ONE wave against itself: the writes of some lanes are read by other
// lanes of the same wave.  LDS operations of a wave execute in issue order, so only the
// compiler has to be stopped from re-ordering them.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// MeanStd.record's inner loop (mean_stds.py:44-48) for one observation feature: float32 running
// sums advanced one worker row at a time, IN ORDER, `sum_sq += v * v` as two rounded operations.
// Rows are read from an LDS tile [rows][stride] sixteen at a time (independent loads in flight)
// so that only the two add chains are serial, not an LDS round trip per row.
__device__ __forceinline__ void record_rows(const float* column, int stride, int rows, float& sum,
                                            float& sum_sq) {
  int w = 0;
  for (; w + 16 <= rows; w += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = column[(w + u) * stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      sum = sum + v[u];
      const float sq = v[u] * v[u];
      sum_sq = sum_sq + sq;
    }
  }
  for (; w < rows; ++w) {
    const float v = column[w * stride];
    sum = sum + v;
    const float sq = v * v;
    sum_sq = sum_sq + sq;
  }
}

// One of the two chains of MeanStd.record on its own: acc += column[w * stride] for w in order.
// Callers stage v and v * v (rounded, as mean_stds.py:47 computes it) side by side and walk the
// two chains on two different waves, i.e. two SIMDs: one dependent float32 add per row and wave
// instead of add + multiply + add — the chain is VALU-issue bound, 4 cycles per instruction.
// The LDS reads of the NEXT 16 rows are issued before the adds of the current 16 (two register
// sets, order pinned): otherwise every batch starts with an exposed LDS round trip, which costs as
// much as the 16 dependent adds themselves.
__device__ __forceinline__ void add_rows(const float* column, int stride, int rows, float& acc) {
  const int full = rows / 16;                       // batches of 16 rows
  float a[16], b[16];
  auto fetch = [&](float (&v)[16], int batch) {
    const int first = 16 * min(batch, max(full - 1, 0));            // past the end: re-read, unused
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = column[(first + u) * stride];
  };
  auto fold = [&](const float (&v)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = acc + v[u];
  };
  if (full > 0) fetch(a, 0);
  for (int batch = 0; batch < full; batch += 2) {
    fetch(b, batch + 1);
    __builtin_amdgcn_sched_barrier(0);
    fold(a);
    __builtin_amdgcn_sched_barrier(0);
    fetch(a, batch + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (batch + 1 < full) fold(b);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int w = 16 * full; w < rows; ++w) acc = acc + column[w * stride];
}

// The same chain over rows that are a CONSTANT 32 floats apart: the row distance is an immediate, so
// two rows come with one ds_read2_b32 — an LDS instruction costs a lone wave ~19 cycles whatever it
// fetches (scripts/ubench/chain.hip), and one per row was 5/6 of the chain's time.
__device__ __forceinline__ void add_rows32(const float* column, int rows, float& acc) {
  constexpr int kPitch = 32;
  const int full = rows / 16;
  float a[16], b[16];
  auto fetch = [&](float (&v)[16], int batch) {
    const float* first = column + 16 * kPitch * min(batch, max(full - 1, 0));      // past the end: re-read
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = first[u * kPitch];
  };
  auto fold = [&](const float (&v)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = acc + v[u];
  };
  if (full > 0) fetch(a, 0);
  for (int batch = 0; batch < full; batch += 2) {
    fetch(b, batch + 1);
    __builtin_amdgcn_sched_barrier(0);
    fold(a);
    __builtin_amdgcn_sched_barrier(0);
    fetch(a, batch + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (batch + 1 < full) fold(b);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int w = 16 * full; w < rows; ++w) acc = acc + column[w * kPitch];
}

// Every 64-byte line of the kernel's argument segment requested at the kernel's entry, as one batch of scalar
// loads.  The latency-chain kernels of the off-policy learner take 1 - 2 KB of arguments; the compiler fetches a
// field where it is first used (there are not enough SGPRs to hold them), and a field on a line nobody has
// touched is a miss of the scalar cache in the middle of a dependent chain — one trip here instead (the entry
// waits for its first arguments anyway), hits afterwards.  The loads' one destination register is dead.
// (each load has a destination of its own, kept alive past the wait: a register the allocator had handed to
//  something else would be overwritten when the load returns)
template <int OFF, typename Args>
__device__ __forceinline__ unsigned kernarg_line(Args args) {
  unsigned sink;
  asm volatile("s_load_dword %0, %1, %2" : "=&s"(sink) : "s"(args), "n"(OFF) : "memory");
  return sink;
}
__device__ __forceinline__ void kernarg_sink(unsigned v) { asm volatile("" :: "s"(v)); }
template <int... LINE>
__device__ __forceinline__ void kernarg_lines(std::integer_sequence<int, LINE...>) {
  auto args = __builtin_amdgcn_kernarg_segment_ptr();
  const unsigned sink[] = {kernarg_line<64 * LINE>(args)...};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  (kernarg_sink(sink[LINE]), ...);
}
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
  kernarg_lines(std::make_integer_sequence<int, (BYTES + 63) / 64>{});
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// MeanStd(clip) as passed through the C ABI (<= 0: no clipping) -> the bound the kernels clamp to.
inline float clip_bound(double norm_clip) {
  return norm_clip > 0 ? (float)norm_clip : __builtin_huge_valf();
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace tonic
