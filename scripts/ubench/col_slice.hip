// Micro-benchmark (round 6, VERDICT r5 item 1 ii): would COLUMN SLICES pay for the off-policy layer phases at
// B = 100?  Today one workgroup carries a 16-row tile through a whole 256 x 256 layer (7 tiles x 4 roles = 28
// workgroups on 256 CUs: 89 % of the chip idle, every phase streams 256 KB of weight image per workgroup).  Sliced:
// S workgroups share a tile, each forms 256 / S output features of the layer and hands its slice of the
// activations to its peers through L2 — "the data is its own flag" (csrc/mlpfwd.h: agent-scope stores into lines
// that hold an empty pattern, agent-scope polls) — then rebuilds the full 16 x 256 row image (row maximum, unit,
// hi / lo split) in its own LDS for the next layer.
//   mode 0  monolithic: one workgroup per tile, whole layer (the shipped decomposition, image pass)
//   mode 1  S = 4 slices per tile, hand-over through L2 every layer
//   mode 2  S = 4 slices, NO hand-over (each workgroup re-uses its own slice: a lower bound — what the slice's
//           stream + MFMAs + epilogue cost without the exchange)
// Per layer and workgroup: mode 0 streams 256 KB (64 blocks of 4 tiles x 8 chunks x 2 terms), 96 MFMAs per wave;
// modes 1 / 2 stream 64 KB, 24 MFMAs per wave (wave w owns ONE 16-feature tile of the slice).
// Build: hipcc -O3 --offload-arch=gfx950 col_slice.hip -o col_slice ; run: ./col_slice [tiles = 7]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_h16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_f16(a, b);
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(b) : "v"(hi), "v"(b));
  lo = pack_f16(a, b);
}

constexpr int kPitch = 256 + 8;                 // halfs per activation row in LDS
constexpr unsigned kEmpty = 0x7fa5c3e1u;        // the exchange lines' "nothing yet" pattern
constexpr int kS = 4;                           // slices per tile

// image: [16 tiles][8 chunks][2 terms][64 lanes][4 dwords]
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* image, int reps, float* xchg, float* sink) {
  __shared__ __attribute__((aligned(16))) _Float16 act[2][16 * kPitch];        // [term][row][k]
  __shared__ float rowmax[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  const int tile = MODE == 0 ? blockIdx.x : blockIdx.x / kS, slice = MODE == 0 ? 0 : blockIdx.x % kS;
  for (int i = threadIdx.x; i < 2 * 16 * kPitch; i += 256) (&act[0][0])[i] = (_Float16)(0.001f * (i & 31));
  __syncthreads();
  constexpr int T = MODE == 0 ? 4 : 1;          // feature tiles per wave
  const u32x4* base[T];
  for (int j = 0; j < T; ++j) {
    const int ft = MODE == 0 ? wave + 4 * j : 4 * slice + wave;
    base[j] = reinterpret_cast<const u32x4*>(image) + (size_t)ft * 8 * 2 * 64 + lane;
  }
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {
    f32x4 acc[T] = {};
    u32x4 wa[T][2], wb[T][2];
    auto fill = [&](u32x4 (&w)[T][2], int c) {
#pragma unroll
      for (int j = 0; j < T; ++j) {
        w[j][0] = base[j][(size_t)((c & 7) * 2 + 0) * 64];
        w[j][1] = base[j][(size_t)((c & 7) * 2 + 1) * 64];
      }
    };
    auto compute = [&](const u32x4 (&w)[T][2], int c) {
      const u32x4 bh = *reinterpret_cast<const u32x4*>(&act[0][m * kPitch + 32 * c + 8 * g]);
      const u32x4 bl = *reinterpret_cast<const u32x4*>(&act[1][m * kPitch + 32 * c + 8 * g]);
#pragma unroll
      for (int j = 0; j < T; ++j) acc[j] = mfma_h16(w[j][1], bh, acc[j]);
#pragma unroll
      for (int j = 0; j < T; ++j) acc[j] = mfma_h16(w[j][0], bl, acc[j]);
#pragma unroll
      for (int j = 0; j < T; ++j) acc[j] = mfma_h16(w[j][0], bh, acc[j]);
    };
    fill(wa, 0);
    for (int c = 0; c < 8; c += 2) {
      fill(wb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(wa, c);
      __builtin_amdgcn_sched_barrier(0);
      fill(wa, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(wb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: bias + ReLU of this lane's features (row m, features 16 ft + 4 g + e)
    float h[T][4];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) h[j][e] = fmaxf(acc[j][e] * 1e-3f + 0.01f * (j + e + 1), 0.f);
    __syncthreads();                            // (everybody has read the old image)
    if (MODE == 1) {
      // publish the slice: the tile's line area of THIS repetition, [row][256 features] floats; one writer per
      // 16-byte group; then gather the whole row image from all four slices (the own one included: uniform code)
      float* area = xchg + ((size_t)r * gridDim.x / kS + tile) * 16 * 256;
      const int f = 64 * slice + 16 * wave + 4 * g;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        __hip_atomic_store(area + m * 256 + f + e, h[0][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // thread (row = tid >> 4, slot = tid & 15) collects features 4 slot + 64 u, u < 4, of its row
      const int prow = threadIdx.x >> 4, slot = threadIdx.x & 15;
      // all 16 words of this thread are requested TOGETHER (one L2 round trip when the peers are done), and
      // requested again while any of them is still empty
      float v[4][4], mx = 0.f;
      bool missing;
      do {
        unsigned bits[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            bits[u][e] = __hip_atomic_load(reinterpret_cast<const unsigned*>(area + prow * 256 + 64 * u + 4 * slot + e),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        missing = false;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            missing |= bits[u][e] == kEmpty;
            v[u][e] = __uint_as_float(bits[u][e]);
          }
        if (missing) __builtin_amdgcn_s_sleep(1);
      } while (missing);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[u][e]);
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 16));
      int ex = __builtin_amdgcn_frexp_expf(mx);
      ex = ex < -38 ? -38 : (ex > 100 ? 100 : ex);
      const float unit = __uint_as_float((unsigned)(127 + 14 - ex) << 23);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned h0, l0, h1, l1;
        split2_pair(v[u][0] * unit, v[u][1] * unit, h0, l0);
        split2_pair(v[u][2] * unit, v[u][3] * unit, h1, l1);
        *reinterpret_cast<u32x2*>(&act[0][prow * kPitch + 64 * u + 4 * slot]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&act[1][prow * kPitch + 64 * u + 4 * slot]) = u32x2{l0, l1};
      }
    } else {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < T; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, h[j][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (g == 0) rowmax[wave][m] = mx;
      __syncthreads();
      mx = fmaxf(fmaxf(rowmax[0][m], rowmax[1][m]), fmaxf(rowmax[2][m], rowmax[3][m]));
      int ex = __builtin_amdgcn_frexp_expf(mx);
      ex = ex < -38 ? -38 : (ex > 100 ? 100 : ex);
      const float unit = __uint_as_float((unsigned)(127 + 14 - ex) << 23);
#pragma unroll
      for (int j = 0; j < T; ++j) {
        unsigned h0, l0, h1, l1;
        split2_pair(h[j][0] * unit, h[j][1] * unit, h0, l0);
        split2_pair(h[j][2] * unit, h[j][3] * unit, h1, l1);
        const int f = MODE == 0 ? 16 * (wave + 4 * j) + 4 * g : 64 * slice + 16 * wave + 4 * g;
        *reinterpret_cast<u32x2*>(&act[0][m * kPitch + f]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&act[1][m * kPitch + f]) = u32x2{l0, l1};
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < T; ++j) total += acc[j][0] + acc[j][3];
  }
  if (total == 1.2345e-30f) sink[0] = total;
}

template <int MODE>
void run(const unsigned* W, float* xchg, size_t xchg_bytes, float* sink, int tiles) {
  const int reps = 100, grid = MODE == 0 ? tiles : tiles * kS;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int trial = 0; trial < 4; ++trial) {
    (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(xchg), kEmpty, xchg_bytes / 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, W, reps, xchg, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (trial > 0 && ms < best) best = ms;
  }
  const char* mode[3] = {"monolithic: 1 workgroup per 16-row tile, whole layer      ",
                         "4 column slices per tile, hand-over through L2 per layer  ",
                         "4 column slices per tile, NO hand-over (lower bound)      "};
  printf("%d tiles  %s %6.2f us per 256 x 256 layer phase (%d workgroups)\n", tiles, mode[MODE], best * 1e3 / reps, grid);
}

int main(int argc, char** argv) {
  unsigned* W; float* sink; float* xchg;
  (void)hipMalloc(&W, 16 * 8 * 2 * 64 * 16 + 4096); (void)hipMalloc(&sink, 64);
  (void)hipMemset(W, 0, 16 * 8 * 2 * 64 * 16 + 4096);
  for (int tiles : {7, 64}) {
    const size_t xchg_bytes = (size_t)100 * tiles * 16 * 256 * 4;
    (void)hipMalloc(&xchg, xchg_bytes);
    run<0>(W, xchg, xchg_bytes, sink, tiles);
    run<1>(W, xchg, xchg_bytes, sink, tiles);
    run<2>(W, xchg, xchg_bytes, sink, tiles);
    (void)hipFree(xchg);
  }
  return 0;
}
