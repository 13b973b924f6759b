// Does the dependent packed-FMA pair the SLP vectoriser put into the critic grad kernel (round 4,
// profiles/r04_determinism.md) give wrong results on gfx950?
//   v_pk_fma_f32 T, X, Y, C  op_sel_hi:[1,0,1]      T.lo = X.lo*Y.lo + C.lo   T.hi = X.hi*Y.lo + C.hi
//   s_nop N
//   v_pk_fma_f32 T, Z, Y, T  op_sel:[0,1,0]         T.lo = Z.lo*Y.hi + T.lo   T.hi = Z.hi*Y.hi + T.hi
// between v_mfma_f32_16x16x4_f32 instructions, two waves per SIMD, against the same arithmetic on
// scalar v_fma_f32 (bit compare).  Variants: the s_nop between the pair (0 as compiled, 1, 3), with
// and without MFMAs around.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mix(unsigned i, unsigned lane, unsigned salt) {
  unsigned h = i * 2654435761u ^ (lane * 40503u + salt * 9176u);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  return __uint_as_float(0x3f000000u | (h & 0x007fffffu)) - 0.75f;        // [-0.25, 0.25)
}

template <int NOP, bool MFMA, bool PRE>
__global__ __launch_bounds__(512) void k(unsigned long long* bad, float* out, int iters) {
  const unsigned lane = threadIdx.x + blockIdx.x * blockDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float ma = 1.0f + (threadIdx.x & 15) * 1e-3f, mb = 0.5f;
  f32x2 t = {0.f, 0.f};
  float rlo = 0.f, rhi = 0.f, sink = 0.f;
  unsigned long long wrong_lo = 0, wrong_hi = 0;
  for (int i = 0; i < iters; ++i) {
    const f32x2 x = {mix(i, lane, 1), mix(i, lane, 2)}, z = {mix(i, lane, 3), mix(i, lane, 4)};
    const f32x2 y = {mix(i, lane, 5), mix(i, lane, 6)};
    const f32x2 c = {rlo, rhi};                          // (the accumulators carried by the reference)
    float p0 = mix(i, lane, 7), p1 = mix(i, lane, 8);
    if (MFMA) {
      if (NOP == 0)
        asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            "v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_fma_f32 %1, %5, %3, %1 op_sel:[0,1,0]\n\t"
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            : "+v"(acc), "+v"(t) : "v"(x), "v"(y), "v"(c), "v"(z), "v"(ma), "v"(mb), "v"(p0), "v"(sink), "v"(p1));
      else if (NOP == 1)
        asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            "v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\t"
            "s_nop 1\n\t"
            "v_pk_fma_f32 %1, %5, %3, %1 op_sel:[0,1,0]\n\t"
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            : "+v"(acc), "+v"(t) : "v"(x), "v"(y), "v"(c), "v"(z), "v"(ma), "v"(mb), "v"(p0), "v"(sink), "v"(p1));
      else
        asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            "v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\t"
            "s_nop 3\n\t"
            "v_pk_fma_f32 %1, %5, %3, %1 op_sel:[0,1,0]\n\t"
            "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\t"
            : "+v"(acc), "+v"(t) : "v"(x), "v"(y), "v"(c), "v"(z), "v"(ma), "v"(mb), "v"(p0), "v"(sink), "v"(p1));
    } else {
      asm volatile(
          "v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\t"
          "s_nop 0\n\t"
          "v_pk_fma_f32 %1, %5, %3, %1 op_sel:[0,1,0]\n\t"
          : "+v"(acc), "+v"(t) : "v"(x), "v"(y), "v"(c), "v"(z), "v"(p0), "v"(sink), "v"(p1));
    }
    (void)PRE;
    const float want_lo = __builtin_fmaf(z[0], y[1], __builtin_fmaf(x[0], y[0], rlo));
    const float want_hi = __builtin_fmaf(z[1], y[1], __builtin_fmaf(x[1], y[0], rhi));
    wrong_lo += __float_as_uint(t[0]) != __float_as_uint(want_lo);
    wrong_hi += __float_as_uint(t[1]) != __float_as_uint(want_hi);
    rlo = want_lo * 0.5f; rhi = want_hi * 0.5f;          // (keep the accumulators bounded)
  }
  if (wrong_lo) atomicAdd(bad, wrong_lo);
  if (wrong_hi) atomicAdd(bad + 1, wrong_hi);
  out[lane] = acc[0] + acc[1] + acc[2] + acc[3] + sink + t[0] + t[1];
}

// The pair with its neighbourhood in the kernel: Z.hi overwritten right behind the pair (the next
// tile's prefetched input moves in), one MFMA, then scalar FMAs consuming T.lo / T.hi — hard registers.
template <int GAP>
__global__ __launch_bounds__(512) void k2(unsigned long long* bad, float* out, int iters) {
  const unsigned lane = threadIdx.x + blockIdx.x * blockDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float ma = 1.0f + (threadIdx.x & 15) * 1e-3f, mb = 0.5f;
  float rlo = 0.f, rhi = 0.f;
  unsigned long long wrong_lo = 0, wrong_hi = 0;
  for (int i = 0; i < iters; ++i) {
    const float x0 = mix(i, lane, 1), x1 = mix(i, lane, 2), z0 = mix(i, lane, 3), z1 = mix(i, lane, 4);
    const float y0 = mix(i, lane, 5), y1 = mix(i, lane, 6), p = mix(i, lane, 7), q = mix(i, lane, 8);
    float t0, t1;
    asm volatile(
        "v_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\tv_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\t"
        "v_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\tv_mov_b32 v48, %9\n\tv_mov_b32 v49, %10\n\t"
        "v_mov_b32 v50, %11\n\tv_mov_b32 v51, %12\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %13, %14, %0\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %13, %14, %0\n\t"
        "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[46:47] op_sel_hi:[1,0,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_fma_f32 v[40:41], v[48:49], v[44:45], v[40:41] op_sel:[0,1,0]\n\t"
        "v_mov_b32 v49, v51\n\t"
        "v_mfma_f32_16x16x4_f32 %0, %13, %14, %0\n\t"
        "v_fma_f32 v40, v50, v51, v40\n\t"
        "v_fma_f32 v41, v50, v51, v41\n\t"
        "v_mov_b32 v50, v45\n\t"
        "s_nop 7\n\t"
        "v_mov_b32 %1, v40\n\tv_mov_b32 %2, v41\n\t"
        : "+v"(acc), "=v"(t0), "=v"(t1)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(rlo), "v"(rhi), "v"(z0), "v"(z1), "v"(p), "v"(q),
          "v"(ma), "v"(mb)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    const float want_lo = __builtin_fmaf(p, q, __builtin_fmaf(z0, y1, __builtin_fmaf(x0, y0, rlo)));
    const float want_hi = __builtin_fmaf(p, q, __builtin_fmaf(z1, y1, __builtin_fmaf(x1, y0, rhi)));
    wrong_lo += __float_as_uint(t0) != __float_as_uint(want_lo);
    wrong_hi += __float_as_uint(t1) != __float_as_uint(want_hi);
    rlo = want_lo * 0.5f; rhi = want_hi * 0.5f;
  }
  if (wrong_lo) atomicAdd(bad, wrong_lo);
  if (wrong_hi) atomicAdd(bad + 1, wrong_hi);
  out[lane] = acc[0] + acc[1] + acc[2] + acc[3] + rlo + rhi;
}

template <int NOP, bool MFMA>
void run(unsigned long long* bad, float* out, int iters, const char* what) {
  (void)hipMemset(bad, 0, 16);
  hipLaunchKernelGGL((k<NOP, MFMA, true>), dim3(256), dim3(512), 0, 0, bad, out, iters);
  unsigned long long host[2] = {0, 0};
  (void)hipMemcpy(host, bad, 16, hipMemcpyDeviceToHost);
  printf("%-44s %12.3e pairs: low lane wrong %llu, high lane wrong %llu\n", what,
         256.0 * 512 * iters, host[0], host[1]);
}

int main() {
  unsigned long long* bad; float* out;
  (void)hipMalloc(&bad, 16); (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 100000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, true>(bad, out, iters, "between MFMAs, s_nop 0 (as compiled)");
    run<1, true>(bad, out, iters, "between MFMAs, s_nop 1");
    run<3, true>(bad, out, iters, "between MFMAs, s_nop 3");
    run<0, false>(bad, out, iters, "no MFMAs, s_nop 0");
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((k2<0>), dim3(256), dim3(512), 0, 0, bad, out, iters);
    unsigned long long host[2] = {0, 0};
    (void)hipMemcpy(host, bad, 16, hipMemcpyDeviceToHost);
    printf("%-44s %12.3e pairs: low lane wrong %llu, high lane wrong %llu\n",
           "pair + v_mov Z.hi + MFMA + scalar consumers", 256.0 * 512 * iters, host[0], host[1]);
  }
  return 0;
}
