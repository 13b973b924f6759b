// Micro-benchmark: what ONE 256 x 256 layer of the fused off-policy passes would cost per workgroup (16 batch
// rows, four waves, one workgroup per CU) with the products on fp16x2 terms read from an operand-order weight
// image — DESIGN.md 4.5 (b).  Per repetition and wave: 4 feature tiles x 8 chunks of K = 32: 64 loads of 16 bytes
// (hi + lo term of a tile's chunk: two contiguous 1 KB blocks), 96 v_mfma_f32_16x16x32_f16, the B operand (the
// previous layer's activations as hi / lo binary16 rows) from LDS.
//   mode 0  loads + MFMAs only (B operand in registers)
//   mode 1  + B operand from LDS (16 ds_read_b128 per wave and layer)
//   mode 2  + the layer's epilogue: bias + ReLU, the row's maximum over its 256 features (lane groups by
//           shuffles, waves through LDS), power-of-two unit, hi / lo split, the next layer's LDS image, barriers
// Reference lines of the same harness (scripts/ubench/mfma_vmem.hip): today's fp32 layer 3.97 us (MFMA 3.75,
// stream 4.11); the fp32 operand-order image 3.77 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
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

constexpr int kPitch = 256 + 8;          // halfs per activation row in LDS (16-byte aligned, bank-shifted)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* image, int reps, float* sink) {
  __shared__ __attribute__((aligned(16))) _Float16 act[2][2][16 * kPitch];     // [buffer][term][row][k]
  __shared__ float rowmax[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 2 * 2 * 16 * kPitch; i += 256) (&act[0][0][0])[i] = (_Float16)(0.001f * (i & 31));
  __syncthreads();
  // image: [tile][chunk][term][64 lanes][4 dwords]; this lane's 16 bytes of a block
  const u32x4* base[4];
  for (int j = 0; j < 4; ++j) base[j] = reinterpret_cast<const u32x4*>(image) + (size_t)(wave + 4 * j) * 8 * 2 * 64 + lane;
  u32x4 breg_h = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, breg_l = {1u, 2u, 3u, 4u};
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {
    const int cur = r & 1;
    f32x4 acc[4] = {};
    u32x4 wa[4][2], wb[4][2];
    auto fill = [&](u32x4 (&w)[4][2], int c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[j][0] = base[j][(size_t)((c & 7) * 2 + 0) * 64];
        w[j][1] = base[j][(size_t)((c & 7) * 2 + 1) * 64];
      }
    };
    auto compute = [&](const u32x4 (&w)[4][2], int c) {
      u32x4 bh = breg_h, bl = breg_l;
      if (MODE >= 1) {
        bh = *reinterpret_cast<const u32x4*>(&act[cur][0][m * kPitch + 32 * c + 8 * g]);
        bl = *reinterpret_cast<const u32x4*>(&act[cur][1][m * kPitch + 32 * c + 8 * g]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma_h16(w[j][1], bh, acc[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma_h16(w[j][0], bl, acc[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma_h16(w[j][0], bh, acc[j]);
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
    if (MODE == 2) {
      // bias + ReLU, row maximum (this lane: row m, features 16 tile + 4 g + r), unit, split, next image
      float h[4][4], mx = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[j][e] = fmaxf(acc[j][e] * 1e-3f + 0.01f * (j + e), 0.f);
          mx = fmaxf(mx, h[j][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (g == 0) rowmax[wave][m] = mx;
      __syncthreads();
      mx = fmaxf(fmaxf(rowmax[0][m], rowmax[1][m]), fmaxf(rowmax[2][m], rowmax[3][m]));
      int ex = __builtin_amdgcn_frexp_expf(mx);
      ex = ex < -38 ? -38 : (ex > 100 ? 100 : ex);
      const float unit = __uint_as_float((unsigned)(127 + 14 - ex) << 23);
      const int nxt = cur ^ 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h0, l0, h1, l1;
        split2_pair(h[j][0] * unit, h[j][1] * unit, h0, l0);
        split2_pair(h[j][2] * unit, h[j][3] * unit, h1, l1);
        const int f = 16 * (wave + 4 * j) + 4 * g;
        *reinterpret_cast<u32x2*>(&act[nxt][0][m * kPitch + f]) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(&act[nxt][1][m * kPitch + f]) = u32x2{l0, l1};
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) total += acc[j][0] + acc[j][3];
  }
  if (total == 1.2345e-30f) sink[0] = total;
}

template <int MODE>
void run(const unsigned* W, float* sink) {
  const int reps = 200;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, W, reps, sink);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, W, reps, sink);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const char* mode[3] = {"loads + 96 fp16 MFMAs, B in registers        ", "+ B operand (hi / lo rows) from LDS           ",
                         "+ epilogue: ReLU, row unit, split, next image "};
  printf("fp16x2 on an operand-order image: %s %6.2f us per 256 x 256 layer and workgroup\n", mode[MODE],
         ms / 5 * 1e3 / reps);
}

int main() {
  unsigned* W; float* sink;
  (void)hipMalloc(&W, 16 * 8 * 2 * 64 * 16 + 4096); (void)hipMalloc(&sink, 64);
  (void)hipMemset(W, 0, 16 * 8 * 2 * 64 * 16 + 4096);
  run<0>(W, sink); run<1>(W, sink); run<2>(W, sink);
  return 0;
}
