// Fused forward + loss + backward of the 2x64-tanh PPO networks on 16-sample MFMA tiles,
// TWO waves per SIMD (8 waves / workgroup, <= 256 registers per wave).  The shipped form (CH = 3
// below) runs layer 1, the 64x64 products and the policy head's forward product as fp16x2 terms on
// v_mfma_f32_16x16x32_f16 / 32x32x16_f16, the remaining products on v_mfma_f32_16x16x4_f32; the
// all-fp32 form (CH = 0) and the bf16x3 forms (CH = 1, 2) are kept as references the tests compare
// with (tonic_set_tuning "grad_variant").  tests/mfma_emulator16f.py mirrors the shipped form's
// index design lane by lane in NumPy.
//
// Same mathematics and the same flat gradient image as mlp64.hip's mlp64_grad_kernel (see the
// reference citations there); what changes is the tiling.  PMC on the 32x32x2 version
// (profiles/r01_pmc_grad.md) showed the MFMA pipe busy only 42 % of the time: with one wave per
// SIMD the VALU phases (tanh, Gaussian head, PPO loss) and the LDS transposes of a tile cannot
// overlap its MFMA phases.  A 16-sample tile halves every per-wave register array
// (activations 16 instead of 32 registers, F-layout operands 4 per feature tile), which lets two
// waves share a SIMD so that one wave's MFMA chain runs under the other's VALU / LDS / HBM work.
//
// Layouts (v_mfma_f32_16x16x4_f32: A lane l -> A[i = l&15][k = l>>4], B lane l -> B[k = l>>4][j = l&15],
// D lane l, reg r -> D[row = 4*(l>>4) + r][col = l&15]):
//   S layout: lane = (sample s = lane&15, group g = lane>>4); register q in [0,16) holds hidden
//             feature feat16(q, g) = 16*(q>>2) + 4*g + (q&3).  Transposed products
//             D[out][sample] = sum_k W[out][k] X^T[k][sample] keep the chains x->h1->h2 and
//             dz2->dh1 in registers: MFMA step q contracts feature feat16(q, g) from group g.
//   F layout: lane = (feature i = lane&15 of a 16-feature tile, g); 4 registers = samples 4g..4g+3.
//             Operands of the weight-gradient MFMAs (contraction over the tile's 16 samples),
//             produced by a transpose through a private LDS tile (row stride 24 floats:
//             conflict-free ds_read_b128).
#include "mlp64.h"
#include "collect16.h"

#ifndef TONIC_COLLECT_SC1
#define TONIC_COLLECT_SC1 0
#endif

namespace tonic {

constexpr int kWaves16 = 8;
// tanh(z) = 1 - 2 / (1 + 2^(2 log2(e) z)): the forward weight / bias images of the two hidden layers
// are staged pre-multiplied by 2 log2(e), so the exponent's argument comes straight out of the MFMA
// chain (one multiply per hidden unit and sample less; the backward image of W2 stays unscaled).
constexpr float kTanhScale = 2.8853900817779268f;

__host__ __device__ constexpr int feat16(int q, int g) { return 16 * (q >> 2) + 4 * g + (q & 3); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// CH selects how the two 64x64 hidden-layer products of a tile (h1 -> z2 and dz2 -> dh1) run:
//   0  fp32 MFMA (v_mfma_f32_16x16x4_f32): 64 instructions of 32 cycles per product;
//   1  bf16x3: every fp32 operand is split EXACTLY into three bf16 terms (hi + mid + lo, 8 + 8 + 8
//      significant bits) and the product is the sum of the six bf16 MFMAs whose terms matter at fp32
//      precision (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi; the dropped terms are below 2^-24 of
//      |a||b|), accumulated in fp32: 48 v_mfma_f32_16x16x32_bf16 of ~18 cycles per product plus
//      5.5 VALU instructions per activation for the split.  Weights are split once, at staging.
//   2  = 1 with dW2 on bf16x3 terms as well (2 x 2 tiles of v_mfma_f32_32x32x16_bf16);
//   3  fp16x2: every fp32 operand, scaled by a power of two into the top of binary16's range, is split
//      into two fp16 terms (hi + lo, 11 + 11 bits and the sign of lo: the split leaves at most
//      2^-23 of the operand — 23 significant bits in the worst case, one short of fp32) and the
//      product is the sum of THREE fp16 MFMAs (lo.hi, hi.lo, hi.hi; the dropped lo.lo is at most
//      2^-22 |a||b|, 2^-24 rms), fp32 accumulation — against float64 the gradient sums are as close as
//      those of the fp32-MFMA build (tests/test_gpu_parity.py, tests/test_fp16x2_arithmetic.py): 24
//      v_mfma_f32_16x16x32_f16 per product and 2 VALU instructions per activation for the split
//      (v_cvt_pk_f16_f32 + v_fma_mix_f32).  dW2 runs the same way on 32x32x16 tiles.  Scales: the
//      weight images by the power of two that puts max |W2| below 2^14 (found at staging), h1 by 2^14
//      (|tanh| <= 1), the backward pass by ONE power of two per wave that only ever shrinks — a tile
//      whose gradients would leave the range first rescales the wave's accumulators (exact: powers
//      of two) — so nothing can overflow binary16 and everything that is summed shares a unit.
//      Layer 1 runs the same way, each SAMPLE's inputs in the unit of its own largest |x|, and so does
//      the forward product of a policy head with more than one action (Lds16::HM); heads with one
//      output (the critic) keep their products as per-lane FMAs.
template <int KS1, int AP, int CH>
struct Lds16 {
  static constexpr int TS = CH ? 20 : 24;                // row stride of the transpose tiles
  static constexpr int kW2 = (CH == 1 || CH == 2) ? 6144 : 4096;   // floats per W2 image
  static constexpr int W1I = 0;                          // [4][KS1][64]       (b32 per step)
  // CH 3: [4 T][2 terms][64 lanes][8 fp16]: lane group g, slot e = input XE g + e (layer 1 on fp16x2
  // terms as well: one K = 32 block holds every input of the fused shapes, O <= 32)
  static constexpr int XE = KS1 == 1 ? 4 : 8;
  static constexpr int kW1 = CH == 3 ? 2048 : 4 * KS1 * 64;
  static constexpr int W2S = W1I + kW1;                  // CH 0: [4][4][64][4] (b128 per 4 steps)
  static constexpr int W2B = W2S + kW2;                  // CH 1: [2][4][3 terms][64][8 bf16]
  static constexpr int B1P = W2B + kW2;                  // [4 groups][16]
  static constexpr int B2P = B1P + 64;
  static constexpr int W3P = B2P + 64;                   // [AP][4 groups][16]
  static constexpr int HC = W3P + AP * 64;               // [8][8] head constants
  static constexpr int NORM = HC + 64;                   // mean[4*KS1], std[4*KS1], CH 3: + 1 / std[4*KS1]
  static constexpr int SC = NORM + 12 * KS1;             // CH 3: {max |W2| bits, 1 / forward scale,
                                                         //        1 / backward weight scale, max |W3|,
                                                         //        1 / (head image scale x 2^14 of h2),
                                                         //        -, 1 / W1 image scale}
  // CH 3, policies with more than one action: the head's two products on fp32 MFMA tiles.  Row
  // 4 g + r of the forward image is action g + 4 r, so lane group g of a tile ends up with the
  // outputs of actions g and g + 4 — the k index those actions have in the backward product — and
  // every group does the loss arithmetic of ITS actions only.
  static constexpr int HSLOTS = (AP + 3) / 4;            // actions per lane group
  static constexpr bool HM = CH == 3 && AP > 1;
  // CH 3, layer 1: W1 is equilibrated by COLUMN with powers of two — column k of the image is
  // W1[:, k] 2^(12 - e_k) (e_k: the exponent of max |W1[:, k]|) and input k enters as x_k 2^(e_k), both
  // exact — so that the per-sample unit below is taken over terms of comparable weight: the reference's
  // actor sees RAW observations (models/actors.py:128-129), features seven decades apart with weights
  // to match are the normal case, and a feature 2^-17 below the sample's largest |x| would otherwise
  // lose its low term to binary16's subnormal grid (1e-4 in z1 instead of 2e-7, tests/test_fp16x2_arithmetic.py).
  static constexpr int CXM = SC + 8;                     // [32] max |W1[:, k]| bits
  static constexpr int CX = CXM + 32;                    // [32] 2^(e_k)
  static constexpr int W3F = CX + 32;                    // [2 m][2 terms][64 lanes][8 fp16]: W3[action(row)][feat16(8m+e, g)] x 2^(14 - w3_exp)
  static constexpr int W3B = W3F + (HM ? 1024 : 0);      // [HSLOTS c][64 lanes][4 T]: W3[g + 4c][16 T + i]
  static constexpr int WAVE0 = (W3B + (HM ? HSLOTS * 256 : 0) + 3) / 4 * 4;
  static constexpr int T_FLOATS = 64 * TS;
  static constexpr int DO_FLOATS = 16 * 16;
  static constexpr int WAVE_FLOATS = 2 * T_FLOATS + DO_FLOATS;
  static constexpr int TOTAL = WAVE0 + kWaves16 * WAVE_FLOATS;
  // the epilogue overlays five gradient images (the bucket's largest: O = 4 KS1, A = AP) on all of this
  static constexpr int FOLD = 5 * ((64 * 4 * KS1 + 64 + 4096 + 64 + 66 * AP + kStatSlots + 63) / 64 * 64);
  static constexpr int BYTES = (TOTAL > FOLD ? TOTAL : FOLD) * 4;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// (bf16(a), bf16(b)) packed into one register, round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// a = hi + mid + lo exactly (each residual is exact in fp32 and has at most 8 significant bits
// more than the next term keeps); the three terms of (a, b) come packed pairwise.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid,
                                            unsigned& lo) {
  hi = pack_bf16(a, b);
  a -= __uint_as_float(hi << 16);
  b -= __uint_as_float(hi & 0xffff0000u);
  mid = pack_bf16(a, b);
  a -= __uint_as_float(mid << 16);
  b -= __uint_as_float(mid & 0xffff0000u);
  lo = pack_bf16(a, b);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int kF16Top = 14;          // scaled operands stay below 2^14 (binary16: 65504)

// (fp16(a), fp16(b)) packed, round to nearest even: v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// a = hi + lo + r, |r| <= 2^-23 |a| for |a| in [2^-1, 2^16) (below that the binary16 subnormal grid
// bounds r absolutely: 2^-25); the residual a - hi is exact in fp32 and comes out of ONE
// v_fma_mix_f32 (fp16 half of `hi` x -1 + a) — the compiler itself emits a conversion and a subtract.
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_f16(a, b);
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(b) : "v"(hi), "v"(b));
  lo = pack_f16(a, b);
}

// c / sd with the division's rounding, given r = fl(1 / sd): q0 = fl(c r) is within 1.5 ulp of the
// quotient, its remainder c - q0 sd comes exactly out of one FMA and the correction lands on the
// correctly rounded quotient (Markstein's final step; sd >= 1e-2 here, nothing under- or overflows).
__device__ __forceinline__ float quotient_by(float c, float sd, float r) {
  const float q0 = c * r;
  return fmaf(fmaf(-q0, sd, c), r, q0);
}

// 2^k as a float (k in [-126, 127])
__device__ __forceinline__ float pow2i(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

// max over the 16 lanes of a DPP row, in every lane of the row (four v_max_f32 with DPP operands)
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_lane<0xB1>(v));     // quad_perm [1, 0, 3, 2]
  v = fmaxf(v, dpp_lane<0x4E>(v));     // quad_perm [2, 3, 0, 1]
  v = fmaxf(v, dpp_lane<0x141>(v));    // row_half_mirror
  v = fmaxf(v, dpp_lane<0x140>(v));    // row_mirror
  return v;
}

template <int KS1, int AP, bool ACTOR, int CH>
__device__ __forceinline__ void stage_weights16(float* lds, const MlpArgs& a) {
  using L = Lds16<KS1, AP, CH>;
  const int tid = threadIdx.x, nth = kWaves16 * 64;
  const int O = a.O, A = a.A;
  const float* W1 = a.params;
  const float* b1 = W1 + 64 * O;
  const float* W2 = b1 + 64;
  const float* b2 = W2 + 64 * 64;
  const float* tail = b2 + 64;
  const float* W3 = ACTOR ? tail + A : tail;
  const float* b3 = W3 + (ACTOR ? A * 64 : 64);
  for (int idx = tid; idx < L::kW1; idx += nth) lds[L::W1I + idx] = 0.f;
  if (CH == 3 && tid == 0) lds[L::SC] = 0.f;
  if (CH == 3 && tid < 32) lds[L::CXM + tid] = 0.f;
  __syncthreads();
  if constexpr (CH == 3) {
    // max |W1[:, k]| per input column: the column's scale in the image (built below)
    for (int gi = tid; gi < 64 * O; gi += nth)
      atomicMax(reinterpret_cast<unsigned*>(lds + L::CXM) + gi % O, __float_as_uint(fabsf(W1[gi])));
  } else {
    for (int gi = tid; gi < 64 * O; gi += nth) {          // coalesced reads, LDS scatter
      const int row = gi / O, k = gi - row * O;
      const int T = row >> 4, i = row & 15, st = k >> 2, gg = k & 3;
      lds[L::W1I + (T * KS1 + st) * 64 + gg * 16 + i] = W1[gi] * kTanhScale;
    }
  }
  int w_exp = 0;   // CH 3: the exponent that bounds every |W2| (|w| < 2^w_exp), the same in every workgroup
  {
    // all eight W2 values of this thread are requested before the first LDS scatter (a load in a
    // plain copy loop is waited for before the next one is issued)
    constexpr int kPer = 64 * 64 / (kWaves16 * 64);
    float w2v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) w2v[u] = W2[tid + u * nth];
    if constexpr (CH == 3) {
      if (tid < 64) {           // max |W3| over the live heads: bounds dz2 from the head gradients
        const int nout = ACTOR ? A : 1;
        float m3 = 0.f;
        for (int aa = 0; aa < nout; ++aa) m3 = fmaxf(m3, fabsf(W3[aa * 64 + tid]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        if (tid == 0) lds[L::SC + 3] = m3;
      }
      float m = 0.f;
#pragma unroll
      for (int u = 0; u < kPer; ++u) m = fmaxf(m, fabsf(w2v[u]));
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if ((tid & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(lds + L::SC), __float_as_uint(m));
      __syncthreads();
      w_exp = __builtin_amdgcn_frexp_expf(lds[L::SC]);
      w_exp = w_exp < -60 ? -60 : (w_exp > 60 ? 60 : w_exp);
      if (tid == 0) {
        lds[L::SC + 1] = pow2i(w_exp - 2 * kF16Top + 2);    // 1 / (forward image scale x 2^14 of h1)
        lds[L::SC + 2] = pow2i(w_exp - kF16Top);            // 1 / backward image scale
      }
      // layer 1's image, column k: |w kTanhScale| < 2^(e_k + 2), scaled by 2^(12 - e_k); the
      // inputs come times 2^(e_k) (Lds16::CX), the products in the unit 2^(2 - 14)
      auto column_exp = [&](int k) {
        const int e = __builtin_amdgcn_frexp_expf(lds[L::CXM + k]);
        return e < -40 ? -40 : (e > 40 ? 40 : e);
      };
      if (tid == 0) lds[L::SC + 6] = pow2i(2 - kF16Top);
      if (tid < 32) lds[L::CX + tid] = pow2i(column_exp(tid));
      unsigned short* img1 = reinterpret_cast<unsigned short*>(lds + L::W1I);
      for (int gi = tid; gi < 64 * O; gi += nth) {
        const int row = gi / O, k = gi - row * O;
        const int T = row >> 4, i = row & 15, gg = k / L::XE, e = k - gg * L::XE;
        unsigned hi, lo;
        split2_pair(W1[gi] * kTanhScale * pow2i(kF16Top - 2 - column_exp(k)), 0.f, hi, lo);
        img1[((T * 2 + 0) * 64 + gg * 16 + i) * 8 + e] = (unsigned short)hi;
        img1[((T * 2 + 1) * 64 + gg * 16 + i) * 8 + e] = (unsigned short)lo;
      }
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int gi = tid + u * nth, row = gi >> 6, col = gi & 63;
      const float w = w2v[u];
      if constexpr (CH == 3) {
        // fp16x2 images, the bf16x3 layout with two terms: [2 m][4 T][2 terms][64 lanes][8 fp16].
        // |w kTanhScale| < 2^(w_exp + 2): the forward image is scaled by 2^(12 - w_exp), the
        // backward one by 2^(14 - w_exp).
        unsigned short* fwd = reinterpret_cast<unsigned short*>(lds + L::W2S);
        unsigned short* bwd = reinterpret_cast<unsigned short*>(lds + L::W2B);
        unsigned hi, lo;
        {
          const int T = row >> 4, i = row & 15;
          const int q = ((col >> 4) << 2) | (col & 3), gg = (col >> 2) & 3;
          const int at = ((((q >> 3) * 4 + T) * 2) * 64 + gg * 16 + i) * 8 + (q & 7);
          split2_pair(w * kTanhScale * pow2i(kF16Top - 2 - w_exp), 0.f, hi, lo);
          fwd[at] = (unsigned short)hi;
          fwd[at + 512] = (unsigned short)lo;
        }
        {
          const int T = col >> 4, i = col & 15;
          const int q = ((row >> 4) << 2) | (row & 3), gg = (row >> 2) & 3;
          const int at = ((((q >> 3) * 4 + T) * 2) * 64 + gg * 16 + i) * 8 + (q & 7);
          split2_pair(w * pow2i(kF16Top - w_exp), 0.f, hi, lo);
          bwd[at] = (unsigned short)hi;
          bwd[at + 512] = (unsigned short)lo;
        }
      } else if (CH == 0) {
        {  // forward image: A row = output feature `row`, k = input feature `col`
          const int T = row >> 4, i = row & 15;
          const int st = ((col >> 4) << 2) | (col & 3), gg = (col >> 2) & 3;
          lds[L::W2S + ((T * 4 + (st >> 2)) * 64 + gg * 16 + i) * 4 + (st & 3)] = w * kTanhScale;
        }
        {  // backward image: A row = input feature `col`, k = output feature `row`
          const int T = col >> 4, i = col & 15;
          const int st = ((row >> 4) << 2) | (row & 3), gg = (row >> 2) & 3;
          lds[L::W2B + ((T * 4 + (st >> 2)) * 64 + gg * 16 + i) * 4 + (st & 3)] = w;
        }
      } else {
        // bf16x3 images: MFMA m in {0, 1} contracts the 32 features feat16(8m + e, g), e < 8 — lane
        // (i, g) holds the 8 bf16 of one term contiguously (ds_read_b128)
        unsigned short* fwd = reinterpret_cast<unsigned short*>(lds + L::W2S);
        unsigned short* bwd = reinterpret_cast<unsigned short*>(lds + L::W2B);
        unsigned hi, mid, lo;
        {
          const int T = row >> 4, i = row & 15;
          const int q = ((col >> 4) << 2) | (col & 3), gg = (col >> 2) & 3;
          const int at = ((((q >> 3) * 4 + T) * 3) * 64 + gg * 16 + i) * 8 + (q & 7);
          split3_pair(w * kTanhScale, 0.f, hi, mid, lo);
          fwd[at] = (unsigned short)hi;
          fwd[at + 512] = (unsigned short)mid;
          fwd[at + 1024] = (unsigned short)lo;
        }
        {
          const int T = col >> 4, i = col & 15;
          const int q = ((row >> 4) << 2) | (row & 3), gg = (row >> 2) & 3;
          const int at = ((((q >> 3) * 4 + T) * 3) * 64 + gg * 16 + i) * 8 + (q & 7);
          split3_pair(w, 0.f, hi, mid, lo);
          bwd[at] = (unsigned short)hi;
          bwd[at + 512] = (unsigned short)mid;
          bwd[at + 1024] = (unsigned short)lo;
        }
      }
    }
  }
  for (int idx = tid; idx < 64; idx += nth) {
    const int g = idx >> 4, q = idx & 15;
    lds[L::B1P + idx] = b1[feat16(q, g)] * kTanhScale;
    // CH 3: the layer-2 chain starts from the bias in the unit of its products
    lds[L::B2P + idx] = b2[feat16(q, g)] * kTanhScale * (CH == 3 ? pow2i(2 * kF16Top - 2 - w_exp) : 1.f);
  }
  if constexpr (L::HM) {
    // head images (behind the barrier of the W2 block: SC + 3 holds max |W3|)
    const int nout = ACTOR ? A : 1;
    int w3_exp = __builtin_amdgcn_frexp_expf(lds[L::SC + 3]);
    w3_exp = w3_exp < -60 ? -60 : (w3_exp > 60 ? 60 : w3_exp);
    if (tid == 0) lds[L::SC + 4] = pow2i(w3_exp - 2 * kF16Top);
    unsigned short* img = reinterpret_cast<unsigned short*>(lds + L::W3F);
    for (int idx = tid; idx < 1024; idx += nth) {
      const int e = idx & 7, l = (idx >> 3) & 63, m = idx >> 9, i = l & 15, gg = l >> 4;
      const int aa = (i >> 2) + 4 * (i & 3);
      const float w = ((i & 3) < L::HSLOTS && aa < nout) ? W3[aa * 64 + feat16(8 * m + e, gg)] : 0.f;
      unsigned hi, lo;
      split2_pair(w * pow2i(kF16Top - w3_exp), 0.f, hi, lo);
      img[((m * 2 + 0) * 64 + l) * 8 + e] = (unsigned short)hi;
      img[((m * 2 + 1) * 64 + l) * 8 + e] = (unsigned short)lo;
    }
    for (int idx = tid; idx < L::HSLOTS * 256; idx += nth) {
      const int T = idx & 3, l = (idx >> 2) & 63, c = idx >> 8, i = l & 15, gg = l >> 4;
      const int aa = gg + 4 * c;
      lds[L::W3B + idx] = aa < nout ? W3[aa * 64 + 16 * T + i] : 0.f;
    }
  }
  for (int idx = tid; idx < AP * 64; idx += nth) {
    const int aa = idx >> 6, g = (idx >> 4) & 3, q = idx & 15;
    const int nout = ACTOR ? A : 1;
    lds[L::W3P + idx] = aa < nout ? W3[aa * 64 + feat16(q, g)] : 0.f;
  }
  for (int idx = tid; idx < 8; idx += nth) {
    float bias = 0.f, sigma = 1.f, half_inv_var = 0.f, logc = 0.f;
    if (ACTOR) {
      if (idx < A) {
        bias = b3[idx];
        const float ls = tail[idx];
        const float sp = ls > 20.f ? ls : log1pf(expf(ls));
        sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);
        half_inv_var = 1.0f / (2.0f * (sigma * sigma));
        logc = logf(sigma) + kLogSqrt2Pi;
      }
    } else if (idx == 0) {
      bias = b3[0];
    }
    lds[L::HC + idx * 8 + 0] = bias;
    lds[L::HC + idx * 8 + 1] = sigma;
    lds[L::HC + idx * 8 + 2] = half_inv_var;
    lds[L::HC + idx * 8 + 3] = logc;
    lds[L::HC + idx * 8 + 4] = 1.0f / sigma;
    lds[L::HC + idx * 8 + 5] = 2.0f * half_inv_var;
    lds[L::HC + idx * 8 + 6] = 0.f;
    lds[L::HC + idx * 8 + 7] = 0.f;
  }
  if (!ACTOR) {
    for (int idx = tid; idx < 4 * KS1; idx += nth) {
      lds[L::NORM + idx] = idx < O ? a.norm_mean[idx] : 0.f;
      // CH 3 also stages 1 / std: the normalisations of a tile (eight per lane) run as multiply +
      // one Newton step on the remainder (quotient_by), which gives the DIVISION's rounding of
      // mean_stds.py:36 in 3 instructions instead of the ~10 of an IEEE division
      const float sd = idx < O ? a.norm_std[idx] : 1.f;
      lds[L::NORM + 4 * KS1 + idx] = sd;
      if (CH == 3) lds[L::NORM + 8 * KS1 + idx] = 1.0f / sd;
    }
  }
}

// v + (the same lane of the other three 16-lane groups): two v_permlane*_swap, no LDS round trip.
__device__ __forceinline__ float sum_groups(float v) {
  const unsigned bits = __float_as_uint(v);
  auto p16 = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
  v = __uint_as_float(p16[0]) + __uint_as_float(p16[1]);
  const unsigned b2 = __float_as_uint(v);
  auto p32 = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
  return __uint_as_float(p32[0]) + __uint_as_float(p32[1]);
}

// max(v, the same lane of the other three 16-lane groups)
__device__ __forceinline__ float max_groups(float v) {
  const unsigned bits = __float_as_uint(v);
  auto p16 = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
  v = fmaxf(__uint_as_float(p16[0]), __uint_as_float(p16[1]));
  const unsigned b2 = __float_as_uint(v);
  auto p32 = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
  return fmaxf(__uint_as_float(p32[0]), __uint_as_float(p32[1]));
}

// OUT_EXP: the result comes out times 2^OUT_EXP (exactly the scaled bits of the unscaled result);
// `in_scale` multiplies the argument first (CH 3: the chain's unit back to 1).
// BIAS: the argument is acc x in_scale + bias (CH 3, layer 1: the per-sample unit of x and the bias in one FMA).
template <int OUT_EXP = 0, bool SCALED_IN = false, bool BIAS = false>
__device__ __forceinline__ void tanh16(f32x4 (&acc)[4], float (&out)[16], float in_scale = 1.f,
                                       const f32x4* bias = nullptr) {
  // acc = 2 log2(e) x (kTanhScale); tanh(x) = 1 - 2 / (1 + e^{2x}): four instructions per element
  // (exp, add, rcp, fma) against seven for the odd-symmetric form; absolute error <= 2e-7 over the
  // whole range (e^{2x} = inf gives exactly 1, e^{2x} = 0 exactly -1).
  float t[16], d[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
    t[q] = __builtin_amdgcn_exp2f(BIAS ? fmaf(acc[q >> 2][q & 3], in_scale, bias[q >> 2][q & 3])
                                  : SCALED_IN ? acc[q >> 2][q & 3] * in_scale : acc[q >> 2][q & 3]);
#pragma unroll
  for (int q = 0; q < 16; ++q) d[q] = __builtin_amdgcn_rcpf(1.f + t[q]);
  constexpr float one = (float)(1 << OUT_EXP);
#pragma unroll
  for (int q = 0; q < 16; ++q) out[q] = fmaf(-2.f * one, d[q], one);
}

// acc[T] (+)= sum over 16 steps of W-image chunk x in[]: 64 in-features, 64 out-features.
__device__ __forceinline__ void chain64(const float* wimg, const float (&in)[16], int lane,
                                        f32x4 (&acc)[4]) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(wimg);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 w[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) w[T] = w4[(T * 4 + c) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int T = 0; T < 4; ++T) acc[T] = mfma16(w[T][e], in[4 * c + e], acc[T]);
    }
  }
}

// The same product on bf16x3 terms (Lds16 CH = 1): per 32-feature block the activations are split
// in registers, then six MFMAs per 16-row output tile, smallest terms first.
__device__ __forceinline__ f32x4 mfma_b16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_b32(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// eight fp32 values (two ds_read_b128) -> their three bf16 terms, element order preserved
__device__ __forceinline__ void split3_x8(const f32x4& a, const f32x4& b, u32x4 (&t)[3]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned h, m, l;
    split3_pair(a[2 * p], a[2 * p + 1], h, m, l);
    t[0][p] = h; t[1][p] = m; t[2][p] = l;
    split3_pair(b[2 * p], b[2 * p + 1], h, m, l);
    t[0][2 + p] = h; t[1][2 + p] = m; t[2][2 + p] = l;
  }
}

__device__ __forceinline__ void chain64_b3(const float* wimg, const float (&in)[16], int lane,
                                           f32x4 (&acc)[4]) {
  const u32x4* w4 = reinterpret_cast<const u32x4*>(wimg);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    u32x4 bh, bm, bl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned h, mi, l;
      split3_pair(in[8 * m + 2 * p], in[8 * m + 2 * p + 1], h, mi, l);
      bh[p] = h; bm[p] = mi; bl[p] = l;
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const u32x4 wh = w4[((m * 4 + T) * 3 + 0) * 64 + lane];
      const u32x4 wm = w4[((m * 4 + T) * 3 + 1) * 64 + lane];
      const u32x4 wl = w4[((m * 4 + T) * 3 + 2) * 64 + lane];
      acc[T] = mfma_b16(wl, bh, acc[T]);
      acc[T] = mfma_b16(wh, bl, acc[T]);
      acc[T] = mfma_b16(wm, bm, acc[T]);
      acc[T] = mfma_b16(wm, bh, acc[T]);
      acc[T] = mfma_b16(wh, bm, acc[T]);
      acc[T] = mfma_b16(wh, bh, acc[T]);
    }
  }
}

// The same product on fp16x2 terms (Lds16 CH = 3); `in` is already in binary16's range.
__device__ __forceinline__ f32x4 mfma_h16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_h32(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// eight fp32 values (two ds_read_b128) -> their two fp16 terms, element order preserved
__device__ __forceinline__ void split2_x8(const f32x4& a, const f32x4& b, u32x4 (&t)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned h, l;
    split2_pair(a[2 * p], a[2 * p + 1], h, l);
    t[0][p] = h; t[1][p] = l;
    split2_pair(b[2 * p], b[2 * p + 1], h, l);
    t[0][2 + p] = h; t[1][2 + p] = l;
  }
}

__device__ __forceinline__ void chain64_f2(const float* wimg, const float (&in)[16], int lane,
                                           f32x4 (&acc)[4]) {
  const u32x4* w4 = reinterpret_cast<const u32x4*>(wimg);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    u32x4 bh, bl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned h, l;
      split2_pair(in[8 * m + 2 * p], in[8 * m + 2 * p + 1], h, l);
      bh[p] = h; bl[p] = l;
    }
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const u32x4 wh = w4[((m * 4 + T) * 2 + 0) * 64 + lane];
      const u32x4 wl = w4[((m * 4 + T) * 2 + 1) * 64 + lane];
      acc[T] = mfma_h16(wl, bh, acc[T]);
      acc[T] = mfma_h16(wh, bl, acc[T]);
      acc[T] = mfma_h16(wh, bh, acc[T]);
    }
  }
}

__device__ __forceinline__ void load_bias16(const float* bimg, int g, f32x4 (&acc)[4]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(bimg + g * 16);
#pragma unroll
  for (int T = 0; T < 4; ++T) acc[T] = p[T];
}

template <int TS>
__device__ __forceinline__ void scatter_S16(float* T, const float (&v)[16], int s, int g) {
#pragma unroll
  for (int q = 0; q < 16; ++q) T[feat16(q, g) * TS + s] = v[q];
}

template <int TS>
__device__ __forceinline__ f32x4 gather_F16(const float* T, int tile, int i, int g) {
  return *reinterpret_cast<const f32x4*>(T + (16 * tile + i) * TS + 4 * g);
}

// PROBE builds (developer tool, tonic_debug_grad16_phases) stamp s_memtime at phase boundaries of
// the tile loop and dump per-phase cycle totals of every wave of workgroup 0.
__device__ __forceinline__ unsigned long long probe_clock() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);      // pin the stamp: nothing may be scheduled across it
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PHASE(k)                                              \
  do {                                                        \
    if (PROBE) {                                              \
      const unsigned long long now__ = probe_clock();         \
      ph[k] += now__ - last;                                  \
      last = now__;                                           \
    }                                                         \
  } while (0)

// FWD (critic only): the forward half alone — values -> a.out1, nothing else is formed or written
// (tonic_value_forward on a whole Segment: the same tile loop, input prefetch and layer arithmetic
// as the regression step that follows it).
template <int KS1, int XT, int XR, int AP, bool ACTOR, bool EXACT, int CH, bool PROBE = false,
          bool FWD = false>
__global__ __launch_bounds__(kWaves16 * 64, 2) void mlp64_grad16_kernel(MlpArgs a) {
  static_assert(!FWD || !ACTOR, "the forward-only form serves the critic");
  using L = Lds16<KS1, AP, CH>;
  constexpr int TS16 = L::TS;
  // CH 2: dW2 as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 on bf16x3 terms — the 16 samples of a tile
  // are exactly one K block; 24 MFMAs of 32 cycles instead of 64 fp32 MFMAs of 32 cycles.  Lane
  // (feature l & 31 of a 32-feature tile, half l >> 5) contributes samples 8 half .. 8 half + 7.
  constexpr bool W2B3 = CH >= 2;
  // CH 3: the three 64x64 products on fp16x2 terms (see Lds16); everything behind the loss gradient
  // runs in the wave's unit 2^(kF16Top - e_run) and is brought back when the accumulators are stored.
  constexpr bool F16 = CH == 3;
  // heads on MFMA (Lds16::HM): NS action slots per lane, slot r of lane group g = action g + 4 r
  constexpr bool HM = L::HM && ACTOR;
  constexpr int NS = HM ? L::HSLOTS : AP;
  // one head output (the critic, one-action policies) in the fp16x2 kernels: dW3 = sum_s dz_s h2_s is a
  // 64-vector — sixteen FMAs per lane and tile instead of sixteen fp32 MFMAs on a 16-row tile with
  // one live row (+ 4 for the bias sum), no h2^T / dO tiles; folded over the sample lanes at the end
  constexpr bool H1 = F16 && AP == 1;
  // LDS reads of head weights kept in flight (4 registers each); the widest bucket has none to spare
  constexpr int kW3Window = KS1 >= 8 ? 2 : 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (a.skip != nullptr && *a.skip != 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 15, g = lane >> 4;
  const int i = s;   // feature index inside a 16-feature tile when the lane acts in F layout
  float* TA = lds + L::WAVE0 + wave * L::WAVE_FLOATS;
  float* TB = TA + L::T_FLOATS;
  float* DO = TB + L::T_FLOATS;
  const int O = a.O;
  const int A = EXACT ? AP : a.A;      // EXACT: the action count is the template bucket

  float adv_mean = 0.f, adv_std = 1.f;
  bool adv_norm = false;
  if (ACTOR) {
    adv_mean = a.adv_stats[0];
    adv_std = a.adv_stats[1];
    adv_norm = a.adv_stats[3] != 0.f;
  }

  // dW1: XT full 16-column tiles on MFMA + XR (<= 4) remainder columns on VALU (O = 17 would
  // otherwise pay a whole padded tile — 16 registers and 16 MFMAs per tile — for one column).
  f32x4 gW2[4][4], gW1[4][XT], gW3[4];
  f32x16 gW2w[2][2];
  float gb2w[2] = {0.f, 0.f};
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gW2w[x][y][r] = 0.f;
    }
  }
  float gW1r[4][XR > 0 ? XR : 1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    gW3[x] = zero4;
#pragma unroll
    for (int y = 0; y < 4; ++y) gW2[x][y] = zero4;
#pragma unroll
    for (int y = 0; y < XT; ++y) gW1[x][y] = zero4;
#pragma unroll
    for (int y = 0; y < (XR > 0 ? XR : 1); ++y) gW1r[x][y] = 0.f;
  }
  float gb1[4] = {0.f, 0.f, 0.f, 0.f}, gb2[4] = {0.f, 0.f, 0.f, 0.f};
  // Column sums of the per-sample head gradients (db3 = sum dz, d loss/d sigma) ride on the dW3
  // MFMAs: one more chain against a B operand of ones (rows 0..7 -> db3[a], rows 8..15 -> dsigma[a]).
  f32x4 gHead = {0.f, 0.f, 0.f, 0.f};
  // HM: those sums per lane instead (slot r of lane group g = action g + 4 r; folded over the 16 sample
  // lanes once, at the end): four adds per tile for four fp32 MFMAs, which no VALU work can hide behind
  float hb[2] = {0.f, 0.f}, hsg[2] = {0.f, 0.f};
  float gW3s[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) gW3s[q] = 0.f;
  float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;

  // CH 3: every gradient accumulator above holds (true sum) x 2^(kF16Top - e_run) [x a constant of
  // its class]; e_run only grows.  A tile announces the exponent that bounds its head gradients
  // (x max |W3|: a bound on its dz2) BEFORE anything of it is scaled, so no operand of an fp16
  // product exceeds 2^kF16Top; two binades of headroom keep the rescales to a handful per launch.
  int e_run = -100;
  float s_run = pow2i(kF16Top + 100);
  auto enter_unit = [&](float bound) {
    const int e = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_frexp_expf(bound));
    if (e > e_run) {
      const int e_new = e + 2 > 110 ? 110 : e + 2;
      const float f = __builtin_ldexpf(1.f, e_run - e_new);
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        gb2w[x] *= f;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
#pragma unroll
          for (int r = 0; r < 16; ++r) gW2w[x][y][r] *= f;
        }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        gb1[x] *= f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gW3[x][r] *= f;
#pragma unroll
          for (int y = 0; y < XT; ++y) gW1[x][y][r] *= f;
        }
#pragma unroll
        for (int y = 0; y < (XR > 0 ? XR : 1); ++y) gW1r[x][y] *= f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) gHead[r] *= f;
#pragma unroll
      for (int r = 0; r < 2; ++r) { hb[r] *= f; hsg[r] *= f; }
#pragma unroll
      for (int q = 0; q < 16; ++q) gW3s[q] *= f;
      e_run = e_new;
      s_run = pow2i(kF16Top - e_new);
    }
  };

  // Experiment knob (tonic_set_tuning "grad_skew", default 0): delays the second-dispatched half of
  // the workgroup.  It was meant to put the two waves of a SIMD into opposite MFMA / VALU phases;
  // micro-benchmarks (scripts/ubench) then showed that fp32 MFMA and VALU instructions never
  // overlap on gfx950, so there is nothing to gain from it.
  if (__builtin_amdgcn_readfirstlane(wave) >= kWaves16 / 2) {
    for (int k = 0; k < a.skew; ++k) __builtin_amdgcn_s_sleep(127);
  }
  // Wave priorities (tonic_set_tuning "grad_prio", timing only — same bits).  The issue arbiter prefers the
  // OLDER of the two waves of a SIMD: the second-dispatched half of a workgroup took 14.4 k cycles per tile
  // against 11.8 k (profiles/r04_grad_phases.txt) and a launch ends with its slow half.  2 (default): the
  // two waves of a SIMD swap priorities 0 / 1 every tile (profiles/r05_grad_knobs.txt: actor 206.4 -> 198.7
  // us, critic 163.5 -> 156.2 us per launch).  Raising the fp16 MFMA chains of a tile two more levels bought
  // the actor another 1.8 % — and 104 bytes of scratch per lane (the branches that pick the level split the
  // loop's most register-hungry blocks: HBM traffic 1.05 -> 1.24 x algorithmic, profiles/r05_traffic_chain_prio.json):
  // not kept.  1: the late half at priority 1 throughout (no effect); a start skew of the late half loses.
  const bool late_half = __builtin_amdgcn_readfirstlane(wave) >= kWaves16 / 2;
  if (a.prio == 1 && late_half) __builtin_amdgcn_s_setprio(1);
  int prio_turn = late_half ? 1 : 0;

  // Per-sample inputs, branch-free: out-of-range lanes read a clamped (valid) address and the
  // value is discarded by a select.  (A load under `if (valid)` becomes an exec-masked branch
  // with its own s_waitcnt vmcnt(0): eight serialised HBM round trips per tile.)
  // CH 3: lane group g holds the inputs XE g + e (the B operand of a 16x16x32 MFMA); else 4 st + g
  constexpr int NX = F16 ? L::XE : KS1;
  struct TileIn {
    float x[NX];
    float act[NS];
    float adv, lp, ret;
  };
  auto load_tile = [&](int64_t t, TileIn& in) {
    const int64_t nsl = t * 16 + s;
    const bool ok = nsl < a.n;
    const int64_t nc = ok ? nsl : a.n - 1;
#pragma unroll
    for (int st = 0; st < NX; ++st) {
      const int k = F16 ? L::XE * g + st : 4 * st + g;
      const int kc = k < O ? k : O - 1;
      in.x[st] = a.obs[nc * O + kc];      // RAW: any arithmetic here would wait for the load now
    }
    in.adv = 0.f; in.lp = 0.f; in.ret = 0.f;
#pragma unroll
    for (int aa = 0; aa < NS; ++aa) in.act[aa] = 0.f;
    if (ACTOR) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int aa = HM ? g + 4 * r : r;
        in.act[r] = a.actions[nc * A + (aa < A ? aa : A - 1)];
      }
      in.adv = a.adv[nc];
      in.lp = a.old_logp[nc];
    } else if (!FWD) {
      in.ret = a.returns[nc];
    }
  };

  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last = PROBE ? probe_clock() : 0ull;
  const int64_t ntiles = (a.n + 15) / 16;
  const int64_t tile_stride = (int64_t)gridDim.x * kWaves16;
  TileIn cur, nxt;
  int64_t tile = (int64_t)blockIdx.x * kWaves16 + wave;
  // critic: the first tile's inputs are on their way from HBM while the weights are staged (the
  // actor kernel, at its register limit, loses more in the tile loop than it gains: 287 -> 295 us)
  if (!ACTOR && tile < ntiles) load_tile(tile, cur);
  stage_weights16<KS1, AP, ACTOR, CH>(lds, a);
  __syncthreads();
  float fwd_unit = 1.f, bwd_unit = 1.f, w3_bound = 0.f, head_unit = 1.f, w1_unit = 1.f;
  if constexpr (F16)
    w1_unit = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lds[L::SC + 6])));
  if constexpr (HM)
    head_unit = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lds[L::SC + 4])));
  if constexpr (F16) {
    fwd_unit = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lds[L::SC + 1])));
    bwd_unit = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lds[L::SC + 2])));
    // |dz2| <= sum_a |dzl[a]| max |W3|, and dzl = 2 (...) in both losses
    w3_bound = 2.f * __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(lds[L::SC + 3])));
  }
  if (ACTOR && tile < ntiles) load_tile(tile, cur);
  if constexpr (HM) {
    DO[128 + lane] = 0.f;              // rows 8 .. 15 of dO^T: never written again
    DO[192 + lane] = 0.f;
    wave_lds_sync();
  }
  for (; tile < ntiles; tile += tile_stride) {
    if (a.prio == 2) {
      if (prio_turn & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      ++prio_turn;
    }
    PHASE(11);                                       // loop overhead / previous tail
    const int64_t ns = tile * 16 + s;
    const bool valid = ns < a.n;
    const bool counted = valid && g == 0;

    // consume the prefetched raw inputs: normalise (critic) and zero the padding with a 0/1
    // multiply (a select would let the compiler sink the load into an exec-masked branch)
    float x[NX];
#pragma unroll
    for (int st = 0; st < NX; ++st) {
      const int k = F16 ? L::XE * g + st : 4 * st + g;
      const int kc = k < O ? k : O - 1;
      float v = cur.x[st];
      if (!ACTOR) {                                     // mean_stds.py:36-38 (clip: +inf = none)
        const float centred = v - lds[L::NORM + kc], sd = lds[L::NORM + 4 * KS1 + kc];
        const float hat = F16 ? quotient_by(centred, sd, lds[L::NORM + 8 * KS1 + kc]) : centred / sd;
        v = __builtin_amdgcn_fmed3f(hat, -a.norm_clip, a.norm_clip);
      }
      x[st] = v * ((valid && k < O) ? 1.f : 0.f);
    }
    float (&in_act)[NS] = cur.act;
    const float in_adv = cur.adv, in_lp = cur.lp, in_ret = cur.ret;
    // next tile's inputs are requested now and consumed one iteration later: ~6k cycles of HBM
    // latency per tile (measured with the phase probes) disappear behind this tile's work
    if (tile + tile_stride < ntiles) load_tile(tile + tile_stride, nxt);

    // ---- forward
    float h1[16], h2[16], z[NS], dzl[NS];
    {
      f32x4 acc[4];
      if constexpr (F16) {
        // layer 1 on fp16x2 terms: the sample's inputs in its own unit 2^(14 - ex) (observations
        // have no bound: the exponent of the largest |x| of the sample, over the four lane groups)
        // (every input first times its column's power of two, Lds16::CX: exact)
        float xc[NX], amax = 0.f;
#pragma unroll
        for (int e = 0; e < NX; ++e) {
          xc[e] = x[e] * lds[L::CX + L::XE * g + e];
          amax = fmaxf(amax, fabsf(xc[e]));
        }
        int ex = __builtin_amdgcn_frexp_expf(max_groups(amax));
        ex = ex < -38 ? -38 : (ex > 100 ? 100 : ex);
        const float sx = pow2i(kF16Top - ex);
        u32x4 bh = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < NX / 2; ++p) {
          unsigned h, l;
          split2_pair(xc[2 * p] * sx, xc[2 * p + 1] * sx, h, l);
          bh[p] = h; bl[p] = l;
        }
        const u32x4* w1 = reinterpret_cast<const u32x4*>(lds + L::W1I);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          const u32x4 wh = w1[(T * 2 + 0) * 64 + lane], wl = w1[(T * 2 + 1) * 64 + lane];
          acc[T] = mfma_h16(wl, bh, zero4);
          acc[T] = mfma_h16(wh, bl, acc[T]);
          acc[T] = mfma_h16(wh, bh, acc[T]);
        }
        f32x4 b1v[4];
        load_bias16(lds + L::B1P, g, b1v);
        PHASE(0);                                    // input loads + layer-1 chain issued
        tanh16<kF16Top, true, true>(acc, h1, w1_unit * pow2i(ex - kF16Top), b1v);   // h1 x 2^14 from here on
      } else {
        load_bias16(lds + L::B1P, g, acc);
#pragma unroll
        for (int st = 0; st < KS1; ++st) {
#pragma unroll
          for (int T = 0; T < 4; ++T)
            acc[T] = mfma16(lds[L::W1I + (T * KS1 + st) * 64 + lane], x[st], acc[T]);
        }
        PHASE(0);                                    // input loads + layer-1 chain issued
        tanh16(acc, h1);
      }
      PHASE(1);
      load_bias16(lds + L::B2P, g, acc);
      if constexpr (CH == 0) chain64(lds + L::W2S, h1, lane, acc);
      else if constexpr (F16) chain64_f2(lds + L::W2S, h1, lane, acc);
      else chain64_b3(lds + L::W2S, h1, lane, acc);
      PHASE(2);
      if constexpr (HM) tanh16<kF16Top, true>(acc, h2, fwd_unit);     // h2 x 2^14 as well: the head's operand
      else if constexpr (F16) tanh16<0, true>(acc, h2, fwd_unit);
      else tanh16(acc, h2);
      PHASE(3);
    }
    if constexpr (HM) {
      // z^T[row][sample] = W3 . h2 on fp16x2 terms, six MFMAs: lane (s, g) receives the rows 4 g + r =
      // actions g + 4 r, in the unit 1 / head_unit
      f32x4 zacc = zero4;
      const u32x4* w3f = reinterpret_cast<const u32x4*>(lds + L::W3F);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 bh, bl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned h, l;
          split2_pair(h2[8 * m + 2 * p], h2[8 * m + 2 * p + 1], h, l);
          bh[p] = h; bl[p] = l;
        }
        const u32x4 wh = w3f[(m * 2 + 0) * 64 + lane], wl = w3f[(m * 2 + 1) * 64 + lane];
        zacc = mfma_h16(wl, bh, zacc);
        zacc = mfma_h16(wh, bl, zacc);
        zacc = mfma_h16(wh, bh, zacc);
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) z[r] = zacc[r];          // x head_unit + bias: with the head constants below
    } else {
      // W3 operands through a rolling window of kW3Window LDS reads in flight: the plain loop
      // (read 16 bytes, wait, four FMAs, next) exposed 24 LDS round trips per tile
      const f32x4* w3p = reinterpret_cast<const f32x4*>(lds + L::W3P) + g * 4;
      constexpr int kReads = AP * 4;
      f32x4 win[kW3Window];
#pragma unroll
      for (int u = 0; u < kW3Window && u < kReads; ++u) win[u] = w3p[(u >> 2) * 16 + (u & 3)];
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < kReads; ++t) {
        const int aa = t >> 2, j = t & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) part = fmaf(h2[4 * j + e], win[t % kW3Window][e], part);
        if (t + kW3Window < kReads)
          win[t % kW3Window] = w3p[((t + kW3Window) >> 2) * 16 + ((t + kW3Window) & 3)];
        if (j == 3) {
          z[aa] = sum_groups(part) + lds[L::HC + aa * 8];
          part = 0.f;
        }
      }
    }

    if constexpr (FWD) {
      if (counted) a.out1[ns] = z[0];
      cur = nxt;
      continue;
    }
    const float cw = counted ? 1.f : 0.f;            // arithmetic mask: no lane branches below
    if (ACTOR) {
      float logp = 0.f, loc[NS], dif[NS], dsg[NS], bsum = 0.f;
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int aa = HM ? g + 4 * r : r;                 // HM: this lane group's actions only
        const f32x4 hc = *reinterpret_cast<const f32x4*>(lds + L::HC + aa * 8);
        loc[r] = tanh_fast(HM ? fmaf(z[r], head_unit, hc[0]) : z[r]);
        const float act = valid ? in_act[r] : loc[r];
        dif[r] = act - loc[r];
        const float term = -(dif[r] * dif[r]) * hc[2] - hc[3];
        logp += ((EXACT && !HM) || aa < A) ? term : 0.f;
        if constexpr (F16) bsum = fmaf(fabsf(dif[r]), hc[2], bsum);      // |dzl| <= |gl| |dif| / var
      }
      if constexpr (HM) {                                  // the other groups' actions
        logp = sum_groups(logp);
        bsum = sum_groups(bsum);
      }
      const float old_lp = valid ? in_lp : logp;
      float adv = valid ? in_adv : 0.f;
      if (adv_norm) adv = (adv - adv_mean) / adv_std;
      const float ratio = __expf(logp - old_lp);
      const float clipped_ratio = fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
      const float surr1 = adv * ratio, surr2 = adv * clipped_ratio;
      const bool outside = ratio > a.clip_hi || ratio < a.clip_lo;
      const bool dead = (ratio > a.clip_hi && adv > 0.f) || (ratio < a.clip_lo && adv < 0.f);
      // (a.plain: StochasticPolicyGradient, d(-adv * logp) / d logp = -adv, nothing is clipped)
      const bool plain = a.plain != 0;
      float gl = (valid && (plain || !dead)) ? -(adv * (plain ? 1.f : ratio)) : 0.f;
      if constexpr (F16) {
        enter_unit(row_max16(fabsf(gl) * bsum * w3_bound));
        gl *= s_run;
      }
      st0 += cw * (plain ? -(adv * logp) : -(surr2 < surr1 ? surr2 : surr1));   // torch.min (actors.py:86): a NaN ratio stays NaN, fminf would drop it
      st1 += cw * (old_lp - logp);
      st2 += (counted && outside && !plain) ? 1.f : 0.f;
      st3 += cw;
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int aa = HM ? g + 4 * r : r;
        const f32x2 hs = *reinterpret_cast<const f32x2*>(lds + L::HC + aa * 8 + 4);   // 1/sigma, 1/var
        const float dloc = gl * dif[r] * hs[1];
        const float live = ((EXACT && !HM) || aa < A) ? 1.f : 0.f;
        dzl[r] = live * dloc * (1.f - loc[r] * loc[r]);
        dsg[r] = live * gl * (dif[r] * dif[r] * hs[1] * hs[0] - hs[0]);
        if constexpr (HM) {                                // dO^T: row = action, 16 samples
          DO[aa * 16 + s] = dzl[r];
          hb[r] += dzl[r];
          hsg[r] += dsg[r];
        }
      }
      if constexpr (H1) {
        hsg[0] += dsg[0];
      } else if constexpr (!HM) {
        if (g == 0) {
          f32x4 lo = zero4, hi = zero4;
#pragma unroll
          for (int aa = 0; aa < AP; ++aa) {
            if (aa < 4) lo[aa] = dsg[aa]; else hi[aa - 4] = dsg[aa];
          }
          reinterpret_cast<f32x4*>(DO + s * 16 + 8)[0] = lo;
          reinterpret_cast<f32x4*>(DO + s * 16 + 8)[1] = hi;
        }
      }
    } else {
      const float err = valid ? z[0] - in_ret : 0.f;
      if constexpr (F16) {
        enter_unit(row_max16(fabsf(err) * w3_bound));
        dzl[0] = 2.f * (err * s_run);
      } else {
        dzl[0] = 2.f * err;
      }
      st0 += cw * err * err;
      st1 += cw * z[0];
      st3 += cw;
      if constexpr (!H1) {
        if (g == 0) {
          reinterpret_cast<f32x4*>(DO + s * 16 + 8)[0] = zero4;
          reinterpret_cast<f32x4*>(DO + s * 16 + 8)[1] = zero4;
        }
      }
    }

    PHASE(4);                                        // head + loss
    // ---- backward
    if constexpr (H1) {
      hb[0] += dzl[0];
#pragma unroll
      for (int q = 0; q < 16; ++q) gW3s[q] = fmaf(dzl[0], h2[q], gW3s[q]);
    } else {
      scatter_S16<TS16>(TA, h2, s, g);                     // h2^T for dW3
    }
    if constexpr (!HM && !H1) {
      if (g == 0) {
        f32x4 lo = zero4, hi = zero4;
#pragma unroll
        for (int aa = 0; aa < AP; ++aa) {
          if (aa < 4) lo[aa] = dzl[aa]; else hi[aa - 4] = dzl[aa];
        }
        reinterpret_cast<f32x4*>(DO + s * 16)[0] = lo;
        reinterpret_cast<f32x4*>(DO + s * 16)[1] = hi;
      }
    }
    if constexpr (HM) {
      // (dz . W3)^T[feature][sample] on 4 NS MFMAs: k = lane group = the slot's action
      f32x4 hacc[4] = {zero4, zero4, zero4, zero4};
      const f32x4* w3b = reinterpret_cast<const f32x4*>(lds + L::W3B);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const f32x4 w = w3b[c * 64 + lane];
#pragma unroll
        for (int T = 0; T < 4; ++T) hacc[T] = mfma16(w[T], dzl[c], hacc[T]);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float y = h2[q];                             // x 2^14
        h2[q] = hacc[q >> 2][q & 3] * fmaf(y * -0x1p-28f, y, 1.f);
      }
    } else {
      const f32x4* w3p = reinterpret_cast<const f32x4*>(lds + L::W3P) + g * 4;
      constexpr int kReads = AP * 4;                 // t -> (j = t / AP, aa = t % AP)
      f32x4 win[kW3Window];
#pragma unroll
      for (int u = 0; u < kW3Window && u < kReads; ++u) win[u] = w3p[(u % AP) * 16 + u / AP];
      f32x4 acc = zero4;
#pragma unroll
      for (int t = 0; t < kReads; ++t) {
        const int j = t / AP, aa = t % AP;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(dzl[aa], win[t % kW3Window][e], acc[e]);
        if (t + kW3Window < kReads)
          win[t % kW3Window] = w3p[((t + kW3Window) % AP) * 16 + (t + kW3Window) / AP];
        if (aa == AP - 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = h2[4 * j + e];
            h2[4 * j + e] = acc[e] * fmaf(-y, y, 1.f);          // tanh' in one instruction
          }
          acc = zero4;
        }
      }
    }
    float (&dz2)[16] = h2;
    scatter_S16<TS16>(TB, dz2, s, g);
    wave_lds_sync();
    PHASE(5);                                        // dz2 + scatters

    f32x4 dacc[4] = {zero4, zero4, zero4, zero4};
    if constexpr (CH == 0) chain64(lds + L::W2B, dz2, lane, dacc);          // dh1 = dz2 . W2 (S layout)
    else if constexpr (F16) chain64_f2(lds + L::W2B, dz2, lane, dacc);
    else chain64_b3(lds + L::W2B, dz2, lane, dacc);
    PHASE(6);

    // dW3[a][f] += dO^T . h2  (MFMA: rows = action index, padded to 16)
    if constexpr (!H1) {
      float aop[4];
      if constexpr (HM) {
        // rows 8 .. 15 of dO^T were zeroed once: those rows of gW3 are not used
        const f32x4 row = *reinterpret_cast<const f32x4*>(DO + i * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) aop[e] = row[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) aop[e] = DO[(4 * g + e) * 16 + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) gHead = mfma16(aop[e], 1.f, gHead);
      }
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const f32x4 hF = gather_F16<TS16>(TA, T, i, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) gW3[T] = mfma16(aop[e], hF[e], gW3[T]);
      }
    }
    f32x4 aF[4];
    u32x4 aT[2][3];
    if constexpr (!W2B3) {
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        aF[T] = gather_F16<TS16>(TB, T, i, g);
        gb2[T] += (aF[T][0] + aF[T][1]) + (aF[T][2] + aF[T][3]);
      }
    } else {
#pragma unroll
      for (int Ti = 0; Ti < 2; ++Ti) {
        const f32x4* row = reinterpret_cast<const f32x4*>(TB + (32 * Ti + (lane & 31)) * TS16 +
                                                          8 * (lane >> 5));
        const f32x4 v0 = row[0], v1 = row[1];
        gb2w[Ti] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        if constexpr (F16) {
          u32x4 t2[2];
          split2_x8(v0, v1, t2);
          aT[Ti][0] = t2[0]; aT[Ti][1] = t2[1];
        } else {
          split3_x8(v0, v1, aT[Ti]);
        }
      }
    }
    PHASE(7);                                        // dW3 + dz2^T gathers
    float dz1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // CH 3: h1 carries 2^14; (h1 x -2^-28) h1 + 1 rounds once, like the plain form
      const float nh = F16 ? h1[q] * -0x1p-28f : -h1[q];
      dz1[q] = dacc[q >> 2][q & 3] * fmaf(nh, h1[q], 1.f);
    }
    wave_lds_sync();
    scatter_S16<TS16>(TA, h1, s, g);
    scatter_S16<TS16>(TB, dz1, s, g);
    wave_lds_sync();
    PHASE(8);                                        // dz1 + scatters

    // dW2[out][in] += dz2^T . h1
    if constexpr (!W2B3) {
#pragma unroll
      for (int Tj = 0; Tj < 4; ++Tj) {
        const f32x4 bF = gather_F16<TS16>(TA, Tj, i, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int Ti = 0; Ti < 4; ++Ti) gW2[Ti][Tj] = mfma16(aF[Ti][e], bF[e], gW2[Ti][Tj]);
        }
      }
    } else {
#pragma unroll
      for (int Tj = 0; Tj < 2; ++Tj) {
        const f32x4* row = reinterpret_cast<const f32x4*>(TA + (32 * Tj + (lane & 31)) * TS16 +
                                                          8 * (lane >> 5));
        if constexpr (F16) {
          u32x4 bT[2];
          split2_x8(row[0], row[1], bT);
#pragma unroll
          for (int Ti = 0; Ti < 2; ++Ti) {
            f32x16 acc = gW2w[Ti][Tj];
            acc = mfma_h32(aT[Ti][1], bT[0], acc);
            acc = mfma_h32(aT[Ti][0], bT[1], acc);
            acc = mfma_h32(aT[Ti][0], bT[0], acc);
            gW2w[Ti][Tj] = acc;
          }
        } else {
          u32x4 bT[3];
          split3_x8(row[0], row[1], bT);
#pragma unroll
          for (int Ti = 0; Ti < 2; ++Ti) {
            f32x16 acc = gW2w[Ti][Tj];
            acc = mfma_b32(aT[Ti][2], bT[0], acc);
            acc = mfma_b32(aT[Ti][0], bT[2], acc);
            acc = mfma_b32(aT[Ti][1], bT[1], acc);
            acc = mfma_b32(aT[Ti][1], bT[0], acc);
            acc = mfma_b32(aT[Ti][0], bT[1], acc);
            acc = mfma_b32(aT[Ti][0], bT[0], acc);
            gW2w[Ti][Tj] = acc;
          }
        }
      }
    }
    PHASE(9);                                        // dW2
    // dW1[out][in] += dz1^T . x
    f32x4 cF[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      cF[T] = gather_F16<TS16>(TB, T, i, g);
      gb1[T] += (cF[T][0] + cF[T][1]) + (cF[T][2] + cF[T][3]);
    }
    wave_lds_sync();
#pragma unroll
    for (int st = 0; st < NX; ++st) TA[((F16 ? L::XE * g + st : 4 * st + g)) * TS16 + s] = x[st];
    wave_lds_sync();
#pragma unroll
    for (int Tj = 0; Tj < XT; ++Tj) {
      f32x4 xF = gather_F16<TS16>(TA, Tj, i, g);
      if (16 * Tj + i >= 4 * KS1) xF = zero4;          // rows beyond the staged x hold stale data
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int Ti = 0; Ti < 4; ++Ti) gW1[Ti][Tj] = mfma16(cF[Ti][e], xF[e], gW1[Ti][Tj]);
      }
    }
#pragma unroll
    for (int c = 0; c < XR; ++c) {                      // remainder columns 16*XT + c on VALU
      const f32x4 xr = *reinterpret_cast<const f32x4*>(TA + (16 * XT + c) * TS16 + 4 * g);
#pragma unroll
      for (int Ti = 0; Ti < 4; ++Ti) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gW1r[Ti][c] = fmaf(cF[Ti][e], xr[e], gW1r[Ti][c]);
      }
    }
    wave_lds_sync();
    cur = nxt;
    PHASE(10);                                       // dW1
  }
  if (PROBE && blockIdx.x == 0 && lane == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out1) + wave * 12;
    for (int k = 0; k < 12; ++k) dst[k] = ph[k];
  }
  if constexpr (FWD) return;

  if constexpr (F16) {
    // back from the wave's unit (powers of two: exact); factor by factor, so that no product of
    // factors leaves the normal range on its own
    const float inv_s = pow2i(e_run - kF16Top);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      gb2w[x] *= inv_s;
#pragma unroll
      for (int y = 0; y < 2; ++y) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gW2w[x][y][r] = gW2w[x][y][r] * inv_s * 0x1p-14f;   // h1's 2^14
      }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      gb1[x] = gb1[x] * inv_s * bwd_unit;                                    // the W2 image's scale
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gW3[x][r] = HM ? gW3[x][r] * inv_s * 0x1p-14f : gW3[x][r] * inv_s;         // HM: h2's 2^14
#pragma unroll
        for (int y = 0; y < XT; ++y) gW1[x][y][r] = gW1[x][y][r] * inv_s * bwd_unit;
      }
#pragma unroll
      for (int y = 0; y < (XR > 0 ? XR : 1); ++y) gW1r[x][y] = gW1r[x][y] * inv_s * bwd_unit;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gHead[r] *= inv_s;
#pragma unroll
    for (int r = 0; r < 2; ++r) { hb[r] *= inv_s; hsg[r] *= inv_s; }
#pragma unroll
    for (int q = 0; q < 16; ++q) gW3s[q] *= inv_s;
  }

  // ---------------- fold into the flat gradient image (same layout as mlp64_grad_kernel)
  // `G[i] += v` per wave in turn would be a dependent LDS read-modify-write per element (the
  // compiler must assume the addresses alias): ~100 serialised round trips per wave and turn,
  // 40 us per launch — and ds_add_f32 is slower still.  Instead four waves at a time STORE their
  // accumulators into four private images (plain, pipelined ds_write), then all 512 threads add
  // the images element-wise onto image 0 in wave order: 0 + w0 + w1 + ... + w7 left to right,
  // bit-identical to the serial fold, in two short rounds.  The images overlay the weight and
  // tile areas, which are dead by now.
  const int oW1 = 0, ob1 = 64 * O, oW2 = ob1 + 64, ob2 = oW2 + 4096, oTail = ob2 + 64;
  const int oLs = oTail, oW3 = ACTOR ? oTail + A : oTail, ob3 = oW3 + (ACTOR ? A * 64 : 64);
  const int nout = ACTOR ? A : 1;
  const int P = ob3 + nout;
  const int image = (P + kStatSlots + 63) / 64 * 64;
  float* G = lds;
  for (int round = 0; round < kWaves16 / 4; ++round) {
    __syncthreads();
    if ((wave >> 2) == round) {
      float* IMG = lds + (1 + (wave & 3)) * image;
#pragma unroll
      for (int Ti = 0; Ti < 4; ++Ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * Ti + 4 * g + r;
          if constexpr (!W2B3) {
#pragma unroll
            for (int Tj = 0; Tj < 4; ++Tj) IMG[oW2 + row * 64 + 16 * Tj + s] = gW2[Ti][Tj][r];
          }
#pragma unroll
          for (int Tj = 0; Tj < XT; ++Tj)
            if (16 * Tj + s < O) IMG[oW1 + row * O + 16 * Tj + s] = gW1[Ti][Tj][r];
        }
        const float v1 = sum_groups(gb1[Ti]);
        if (g == 0) IMG[ob1 + 16 * Ti + i] = v1;
        if constexpr (!W2B3) {
          const float v2 = sum_groups(gb2[Ti]);
          if (g == 0) IMG[ob2 + 16 * Ti + i] = v2;
        }
#pragma unroll
        for (int c = 0; c < XR; ++c) {                  // lane (i, g): feature 16*Ti + i, 4 samples
          const float vr = sum_groups(gW1r[Ti][c]);
          if (g == 0 && 16 * XT + c < O) IMG[oW1 + (16 * Ti + i) * O + 16 * XT + c] = vr;
        }
        if constexpr (!H1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int aa = 4 * g + r;                    // gW3[T] rows are action indices
            if (aa < nout) IMG[oW3 + aa * 64 + 16 * Ti + s] = gW3[Ti][r];
          }
        }
      }
      if constexpr (W2B3) {
        // 32x32 tiles: lane l, register r -> row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
#pragma unroll
        for (int Ti = 0; Ti < 2; ++Ti) {
#pragma unroll
          for (int Tj = 0; Tj < 2; ++Tj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = 32 * Ti + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
              IMG[oW2 + row * 64 + 32 * Tj + (lane & 31)] = gW2w[Ti][Tj][r];
            }
          }
          const unsigned bits = __float_as_uint(gb2w[Ti]);
          auto halves = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
          const float v2 = __uint_as_float(halves[0]) + __uint_as_float(halves[1]);
          if (lane < 32) IMG[ob2 + 32 * Ti + lane] = v2;
        }
      }
      if constexpr (H1) {
        // per-lane sums (feature feat16(q, g), this lane's samples) -> over the 16 sample lanes of the row
        auto row_sum = [](float v) {
          v += dpp_lane<0xB1>(v);
          v += dpp_lane<0x4E>(v);
          v += dpp_lane<0x141>(v);
          v += dpp_lane<0x140>(v);
          return v;
        };
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = row_sum(gW3s[q]);
          if (s == 0) IMG[oW3 + feat16(q, g)] = v;
        }
        const float vb = row_sum(hb[0]), vs = row_sum(hsg[0]);
        if (lane == 0) {
          IMG[ob3] = vb;
          if (ACTOR) IMG[oLs] = vs;
        }
      } else if constexpr (HM) {
        // per-lane head sums -> over the 16 samples of the lane's row, then lane s = 0 of group g
        // stores the sums of its actions g + 4 r
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          float vb = hb[r], vs = hsg[r];
          vb += dpp_lane<0xB1>(vb); vs += dpp_lane<0xB1>(vs);        // quad_perm [1, 0, 3, 2]
          vb += dpp_lane<0x4E>(vb); vs += dpp_lane<0x4E>(vs);        // quad_perm [2, 3, 0, 1]
          vb += dpp_lane<0x141>(vb); vs += dpp_lane<0x141>(vs);      // row_half_mirror
          vb += dpp_lane<0x140>(vb); vs += dpp_lane<0x140>(vs);      // row_mirror
          const int aa = g + 4 * r;
          if (s == 0 && aa < nout) { IMG[ob3 + aa] = vb; IMG[oLs + aa] = vs; }
        }
      } else if (s == 0) {
        // gHead rows: lane (j, g), reg r -> row 4g + r (every column j holds the same sum)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * g + r;
          if (row < 8 && row < nout) IMG[ob3 + row] = gHead[r];
          if (ACTOR && row >= 8 && row - 8 < nout) IMG[oLs + row - 8] = gHead[r];
        }
      }
      const float r0 = wave_sum(st0), r1 = wave_sum(st1), r2 = wave_sum(st2), r3 = wave_sum(st3);
      if (lane == 0) { IMG[P + 0] = r0; IMG[P + 1] = r1; IMG[P + 2] = r2; IMG[P + 5] = r3; }
      if (lane == 0) { IMG[P + 3] = 0.f; IMG[P + 4] = 0.f; IMG[P + 6] = 0.f; IMG[P + 7] = 0.f; }
    }
    __syncthreads();
    for (int idx = tid; idx < P + kStatSlots; idx += kWaves16 * 64) {
      float acc = round == 0 ? 0.f : G[idx];
#pragma unroll
      for (int w = 1; w <= 4; ++w) acc += lds[w * image + idx];
      G[idx] = acc;
    }
  }
  __syncthreads();
  float* dst = a.out0 + (int64_t)blockIdx.x * a.pstride;
  for (int idx = tid; idx < P + kStatSlots; idx += kWaves16 * 64) dst[idx] = G[idx];
}

// ------------------------------------------------- packed-weight acting (device-resident collect)
//
// The per-environment-step collect kernel is latency-bound (W = 256 observations = 16 tiles):
// staging 21 KB of weights through LDS at every launch cost more than the MFMA chain itself.
// `actor_pack_kernel` therefore writes the pre-permuted operand images ONCE per learner update
// into HBM (L2-resident afterwards) and `ppo_collect16_kernel` streams its MFMA A operands
// straight from that image (lane-linear, coalesced) — no weight staging through LDS.
__global__ void actor_pack_kernel(const float* params, float* packed, int O, int A, int ks1,
                                  int ap) {
  const PackedActor L(ks1, ap);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  const float* W1 = params;
  const float* b1 = W1 + 64 * O;
  const float* W2 = b1 + 64;
  const float* b2 = W2 + 64 * 64;
  const float* tail = b2 + 64;
  const float* W3 = tail + A;
  const float* b3 = W3 + A * 64;
  for (int idx = tid; idx < 4 * ks1 * 64; idx += nth) {
    const int T = idx / (ks1 * 64), rem = idx - T * ks1 * 64, st = rem >> 6, l = rem & 63;
    const int k = 4 * st + (l >> 4);
    packed[L.W1I + idx] = k < O ? W1[(16 * T + (l & 15)) * O + k] : 0.f;
  }
  for (int idx = tid; idx < 4096; idx += nth) {
    const int e = idx & 3, l = (idx >> 2) & 63, c = (idx >> 8) & 3, T = idx >> 10;
    packed[L.W2S + idx] = W2[(16 * T + (l & 15)) * 64 + feat16(4 * c + e, l >> 4)];
  }
  for (int idx = tid; idx < 64; idx += nth) {
    packed[L.B1P + idx] = b1[feat16(idx & 15, idx >> 4)];
    packed[L.B2P + idx] = b2[feat16(idx & 15, idx >> 4)];
  }
  for (int idx = tid; idx < ap * 64; idx += nth) {
    const int aa = idx >> 6, g = (idx >> 4) & 3, q = idx & 15;
    packed[L.W3P + idx] = aa < A ? W3[aa * 64 + feat16(q, g)] : 0.f;
  }
  for (int idx = tid; idx < 8; idx += nth) {
    float bias = 0.f, sigma = 1.f, half_inv_var = 0.f, logc = 0.f;
    if (idx < A) {
      bias = b3[idx];
      const float ls = tail[idx];
      const float sp = ls > 20.f ? ls : log1pf(expf(ls));
      sigma = fminf(fmaxf(sp + 1e-8f, 1e-4f), 1.0f);
      half_inv_var = 1.0f / (2.0f * (sigma * sigma));
      logc = logf(sigma) + kLogSqrt2Pi;
    }
    float* hc = packed + L.HC + idx * 8;
    hc[0] = bias; hc[1] = sigma; hc[2] = half_inv_var; hc[3] = logc;
    hc[4] = 1.f / sigma; hc[5] = 2.f * half_inv_var; hc[6] = 0.f; hc[7] = 0.f;
  }
}

// Loads whose values are needed by nobody: the addresses are touched and the results only fold
// into `sink`, which `retire_touches` consumes at the END of the workgroup's work (a store that
// never executes for real data) — so the loads stay in flight behind everything else instead of
// being waited for one by one.
__device__ __forceinline__ void touch(const float* p, float& sink) {
  sink += *p;            // a plain load: it must ALLOCATE in L2 (a nontemporal one streams through)
}
__device__ __forceinline__ void retire_touches(float sink, float* never_written) {
  if (sink == 1.2345e-38f) *never_written = sink;
}

// End of a workgroup's role when the host is waiting on this launch (Collect16Args::done_flags):
// one word per workgroup in pinned host memory, stored at system scope once the workgroup's loads
// from the pinned block have returned (s_waitcnt vmcnt(0)) and — actor role — its actions have
// been released to the system (the fence above).  No cross-workgroup counter: a last-arriver
// protocol costs every workgroup a system-scope fence and an atomic round trip.
__device__ __forceinline__ void collect_signal_done(const Collect16Args& c, int vb) {
  if (c.done_flags == nullptr) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's loads / stores are done
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(c.done_flags + vb, c.done_seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// Actor role, many workers: this workgroup's rows of the Segment's observation row are released
// (the system-scope release in front of the completion word wrote them back) — tell the record role.
__device__ __forceinline__ void collect_signal_rows(const Collect16Args& c, int vb) {
  if (c.tile_done != nullptr && threadIdx.x == 0)
    __hip_atomic_store(c.tile_done + vb, c.done_seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void collect_stamp(const Collect16Args& c, int role, int phase) {
  if (c.stamps != nullptr && threadIdx.x == 0) {
    // (no-return atomics: a read-modify-write would park the wave for an L2 round trip per stamp)
    __hip_atomic_fetch_add(c.stamps + role * 8 + phase, wall_clock64() - c.stamp_t0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (phase == 0)
      __hip_atomic_fetch_add(c.stamps + role * 8 + 7, 1ull, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

constexpr int kCollectLds = 16384;     // floats: MeanStd.record staging tile of the last block
constexpr int kCollectCopyBlocks = 4;  // workgroups that copy the transition outcome

// Block-wide copy src -> dst (and optionally -> lds) of `count` floats whose source is 16-byte
// aligned, as 16-byte requests with ALL of a thread's loads in flight before the first store.
// For sources in pinned HOST memory every load instruction becomes PCIe read requests and every
// dependent batch a PCIe round trip (~2 us): the fields are therefore read exactly once, in as
// few and as wide requests as possible.  Handles up to 8 * threads * 4 floats per call.
// Loads from the pinned host block.  SYS = false (one launch per environment step): streaming
// loads — the launch's acquire made L2 forget the previous step's lines.  SYS = true (resident
// kernel): system-scope loads (sc0 sc1) that bypass the XCD's L2, where the lines of the previous
// step would otherwise still be found.
template <bool SYS>
__device__ __forceinline__ f32x4 host_load4(const float* base, int64_t vec_index) {
  if constexpr (SYS) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                                                 0x7fffffff, 0x27000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(vec_index * 16), 0, 17));
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + vec_index);
  }
}
template <bool SYS>
__device__ __forceinline__ float host_load1(const float* base, int64_t index) {
  if constexpr (SYS) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                                                 0x7fffffff, 0x27000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(index * 4), 0, 17));
  } else {
    return __builtin_nontemporal_load(base + index);
  }
}

template <bool SYS>
__device__ __forceinline__ void wide_copy(const float* src, float* dst, float* lds, int64_t count,
                                          int tid, int threads) {
  const int64_t vecs = count >> 2;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int64_t i = tid + (int64_t)u * threads;
    if (i < vecs) v[u] = host_load4<SYS>(src, i);
  }
  float tail = 0.f;
  const int64_t ti = (vecs << 2) + tid;
  if (ti < count) tail = host_load1<SYS>(src, ti);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int64_t i = tid + (int64_t)u * threads;
    if (i < vecs) {
      if (dst != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * i + e] = v[u][e];       // (dst rows may be 4-byte aligned)
      }
      if (lds != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[4 * i + e] = v[u][e];
      }
    }
  }
  if (ti < count) {
    if (dst != nullptr) dst[ti] = tail;
    if (lds != nullptr) lds[ti] = tail;
  }
}

// HOST: the step inputs (observations, noise, previous outcome) live in pinned host memory that
// the kernel reads in place over PCIe (pinned-host collector, transport 0).
// `vb` of `nb`: the workgroup SLOT whose work this is — the launching grid's own index and size for
// the one-launch-per-step forms; the resident form also runs the slots of workgroups that are not
// there (yet).
template <int KS1, int AP, bool HOST, bool SYS>
__device__ __forceinline__ void collect16_step(const Collect16Args& c, float* tile, int vb, int nb) {
  const int64_t W = c.W;
  const int O = c.O, A = c.A;
  const int tid = threadIdx.x;
  // Workgroup roles: [0, act_blocks) actor tiles | kCollectCopyBlocks outcome-copy blocks | one
  // MeanStd.record block.  The three run side by side on different CUs, so a step costs the
  // longest of {actor chain, cold outcome copy, sequential record} instead of their sum.
  const int act_blocks = nb - 1 - kCollectCopyBlocks;
  if (vb >= act_blocks && vb < act_blocks + kCollectCopyBlocks) {
    // transition outcome (segments.py:27-36): next observations, rewards, resets, terminations
    const int64_t part = vb - act_blocks, stride = 256 * kCollectCopyBlocks;
    float sink = 0.f;
    if (c.pf_next_obs != nullptr) {               // one touch per 64-byte line of the next step
      for (int64_t i = (part * 256 + tid) * 16; i < W * O; i += stride * 16)
        touch(c.pf_next_obs + i, sink);
      for (int64_t i = (part * 256 + tid) * 16; i < W; i += stride * 16) {
        touch(c.pf_rewards + i, sink);
        touch(c.pf_resets + i, sink);
        touch(c.pf_terminations + i, sink);
      }
    }
    if (HOST && c.outcome_row >= 0) {
      // one PCIe round trip per 32 KB: each copy block takes a contiguous, 16-byte aligned
      // quarter of the next observations
      const int64_t total = W * O;
      const int64_t each = ((total + kCollectCopyBlocks - 1) / kCollectCopyBlocks + 3) & ~(int64_t)3;
      const int64_t f0 = min(total, part * each), f1 = min(total, f0 + each);
      float* dst = c.seg_next + c.outcome_row * total;
      // the scalar fields of this block's workers FIRST (their loads then share the PCIe round
      // trip of the observation rows instead of paying a second one behind them)
      const int64_t i0 = part * 256 + tid;
      float rew[2], rst[2], term[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < W) {
          rew[u] = host_load1<SYS>(c.rewards, i);
          rst[u] = host_load1<SYS>(c.resets, i);
          term[u] = host_load1<SYS>(c.terminations, i);
        }
      }
      // (carry-over rows come from the actor tiles, see Collect16Args::next_from_obs: only the
      //  rows of workers that reset are fetched here, behind their flags)
      auto reset_row = [&](int64_t i) {
        for (int k = 0; k < O; ++k) dst[i * O + k] = host_load1<SYS>(c.next_obs, i * O + k);
      };
      if (!c.next_from_obs) {
        for (int64_t f = f0; f < f1; f += 8 * 256 * 4)
          wide_copy<SYS>(c.next_obs + f, dst + f, nullptr, min<int64_t>(8 * 256 * 4, f1 - f), tid, 256);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < W) {
          c.seg_rew[c.outcome_row * W + i] = rew[u];
          c.seg_rst[c.outcome_row * W + i] = rst[u];
          c.seg_term[c.outcome_row * W + i] = term[u];
          if (c.next_from_obs && rst[u] != 0.f) reset_row(i);
        }
      }
      for (int64_t i = i0 + 2 * stride; i < W; i += stride) {          // (W > 2048)
        const float r = host_load1<SYS>(c.rewards, i), x = host_load1<SYS>(c.resets, i),
                    t = host_load1<SYS>(c.terminations, i);
        c.seg_rew[c.outcome_row * W + i] = r;
        c.seg_rst[c.outcome_row * W + i] = x;
        c.seg_term[c.outcome_row * W + i] = t;
        if (c.next_from_obs && x != 0.f) reset_row(i);
      }
    } else if (c.outcome_row >= 0) {
      // all of this thread's loads first, then the stores (a plain copy loop waits per element)
      float* dst = c.seg_next + c.outcome_row * W * O;
      int64_t i = part * 256 + tid;
      for (; i + 7 * stride < W * O; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c.next_obs[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[i + u * stride] = v[u];
      }
      for (; i < W * O; i += stride) dst[i] = c.next_obs[i];
      for (i = part * 256 + tid; i < W; i += stride) {
        const float rew = c.rewards[i], rst = c.resets[i], term = c.terminations[i];
        c.seg_rew[c.outcome_row * W + i] = rew;
        c.seg_rst[c.outcome_row * W + i] = rst;
        c.seg_term[c.outcome_row * W + i] = term;
      }
    }
    retire_touches(sink, c.seg_next);
    if (part == 0) collect_stamp(c, 2, 0);           // copies issued
    collect_signal_done(c, vb);
    if (part == 0) collect_stamp(c, 2, 1);           // flag out
    return;
  }
  if (vb == nb - 1) {
    // MeanStd.record (mean_stds.py:44-48): values and their squares staged side by side, the sum
    // chain on wave 0 and the sum-of-squares chain on wave 1
    if (c.norm_acc == nullptr) { collect_signal_done(c, vb); return; }
    constexpr int kHalf = kCollectLds / 2;
    const int lane = tid & 63, wave = tid >> 6;
    const bool from_segment = HOST && c.tile_done != nullptr;      // scalar (see Collect16Args)
    // The next launch's record reads this step's next observations (trainer.py:44-56 hands them
    // back as the observations of step t + 1): touch them now, from the workgroup slot that will
    // need them, so that the sequential chain does not start behind an HBM round trip.
    float sink = 0.f;
    if (c.pf_next_obs != nullptr || c.outcome_row == c.row)      // (device-resident callers only)
      for (int64_t i = (int64_t)tid * 16; i < W * O; i += 256 * 16) touch(c.next_obs + i, sink);
    float acc = 0.f;
    const float* acc_in = c.norm_acc + c.row * c.norm_stride;
    float* acc_out = c.norm_acc + (c.row + 1) * c.norm_stride;
    // (HOST: the entry was written by whoever ran this slot for the previous step — in the resident
    //  kernel possibly another workgroup on another XCD: agent-scope accesses, coherent per location)
    if (wave < 2 && lane < O) {
      if constexpr (HOST)
        acc = __hip_atomic_load(acc_in + wave * O + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        acc = acc_in[wave * O + lane];
    }
    const int64_t rows_per_chunk = (kHalf / O) & ~3;            // (x4 rows: 16-byte aligned chunks)
    if constexpr (HOST) {
      // A chunk (<= 32 KB) is ONE batch of 16-byte requests, all in flight — one PCIe round trip, or
      // (from_segment) one trip to the Segment row the actor tiles have just released, with the
      // same system-scope loads: they do not trust an L2 line this resident kernel may have touched
      // before the row was written.  TWO chunks are in flight at any time (two register sets): the
      // trip of chunk i + 2 runs under the staging and the chains of chunks i and i + 1; waves 0
      // and 1 request AFTER their chain, waves 2 and 3 at once.
      // Staged at a row pitch of 32 floats whatever O is (<= 32 here): the chain then reads two rows
      // per LDS instruction (add_rows32).
      constexpr int kPitch = 32;
      constexpr int64_t rows_per_chunk = kHalf / kPitch;           // 256 rows, <= 32 KB of requests
      const unsigned row_magic = 0xffffffffu / (unsigned)O + 1u;   // floor(e / O) = umulhi(e, magic), e < 2^27
      const bool whole = (O & 3) == 0;                             // a 16-byte request stays inside a row
      const float* base = from_segment ? c.seg_obs + c.row * W * O : c.obs;
      const int64_t chunks = (W + rows_per_chunk - 1) / rows_per_chunk;
      if (from_segment) {
        // every actor workgroup's rows of this step, ONE poller per word (they all finish within a
        // microsecond or two of each other: their inputs cross PCIe together).  Bounded like every
        // wait of the collector — 1 s, or the resident kernel's park notice: the slots waited for may
        // belong to workgroups that have left — and a row that never came is NOT read: this slot then
        // leaves without its completion word; the host starts the kernel again on the notice, or
        // tonic_collector_wait_actions reports TONIC_ERR_TIMEOUT (naming the word) — never statistics
        // built from a stale row.
        const unsigned long long t0 = wall_clock64();
        bool never = false;
        for (int b = tid; b < act_blocks; b += 256) {
          while (__hip_atomic_load(c.tile_done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                 c.done_seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 100000000ull ||
                (c.abandon != nullptr &&
                 __hip_atomic_load(c.abandon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              never = true;
              break;
            }
          }
        }
        if (__syncthreads_or(never ? 1 : 0)) return;
      }
      collect_stamp(c, 1, 2);                        // (probe) the actor tiles' rows are released
      struct Set { f32x4 v[8]; float tail; };
      auto request = [&](Set& set, int64_t chunk) {
        if (chunk >= chunks) return;                               // scalar
        const int64_t w0 = chunk * rows_per_chunk, rows = min(rows_per_chunk, W - w0);
        const int64_t first = w0 * O, count = rows * O, vecs = count >> 2;
        const int64_t ti = (vecs << 2) + tid;
        const float* src = base + first;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (tid + u * 256 < vecs) set.v[u] = host_load4<SYS>(src, tid + u * 256);
        set.tail = 0.f;
        if (ti < count) set.tail = host_load1<SYS>(src, ti);
      };
      auto stage = [&](const Set& set, int64_t chunk) {
        const int64_t w0 = chunk * rows_per_chunk, rows = min(rows_per_chunk, W - w0);
        const int64_t count = rows * O, vecs = count >> 2;
        auto put = [&](unsigned e, float v) {                      // element e of the chunk
          const unsigned r = O == 1 ? e : __umulhi(e, row_magic);
          const unsigned at = r * kPitch + (e - r * (unsigned)O);
          tile[at] = v;
          tile[kHalf + at] = v * v;
        };
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (tid + u * 256 < vecs) {
            const unsigned e = 4u * (unsigned)(tid + u * 256);
            const f32x4 v = set.v[u];
            if (whole) {
              // (one 16-byte write each: four dword writes at a 16-byte lane stride are 8-way bank
              //  conflicts, 2 us per chunk)
              const unsigned r = __umulhi(e, row_magic);
              const unsigned at = r * kPitch + (e - r * (unsigned)O);
              *reinterpret_cast<f32x4*>(tile + at) = v;
              *reinterpret_cast<f32x4*>(tile + kHalf + at) =
                  f32x4{v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]};
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) put(e + j, v[j]);
            }
          }
        const int64_t ti = (vecs << 2) + tid;
        if (ti < count) put((unsigned)ti, set.tail);
      };
      auto turn = [&](Set& set, int64_t chunk) {
        if (chunk >= chunks) return;                               // scalar
        const int64_t rows = min(rows_per_chunk, W - chunk * rows_per_chunk);
        __syncthreads();                                           // the previous chain has left the tile
        if (chunk == 1) collect_stamp(c, 1, 5);                    // (probe) first chain over
        stage(set, chunk);
        if (chunk == 0) collect_stamp(c, 1, 3);                    // (probe) first chunk arrived, staged
        __syncthreads();
        if (chunk == 0) collect_stamp(c, 1, 4);
        if (wave >= 2) {
          request(set, chunk + 2);
        } else {
          if (lane < O) add_rows32(tile + wave * kHalf + lane, (int)rows, acc);
          request(set, chunk + 2);
        }
      };
      Set even, odd;
      request(even, 0);
      request(odd, 1);
      for (int64_t chunk = 0; chunk < chunks; chunk += 2) {
        turn(even, chunk);
        turn(odd, chunk + 1);
      }
    } else {
      for (int64_t w0 = 0; w0 < W; w0 += rows_per_chunk) {
        const int64_t rows = min(rows_per_chunk, W - w0);
        __syncthreads();
        // staging with eight loads in flight per thread (a plain loop waits for every load)
        const float* src = c.obs + w0 * O;
        const int64_t count = rows * O;
        int64_t i = tid;
        for (; i + 7 * 256 < count; i += 8 * 256) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = src[i + u * 256];
#pragma unroll
          for (int u = 0; u < 8; ++u) { tile[i + u * 256] = v[u]; tile[kHalf + i + u * 256] = v[u] * v[u]; }
        }
        for (; i < count; i += 256) {
          const float v = src[i];
          tile[i] = v;
          tile[kHalf + i] = v * v;
        }
        __syncthreads();
        if (wave < 2 && lane < O) add_rows(tile + wave * kHalf + lane, O, (int)rows, acc);
      }
    }
    if (wave < 2 && lane < O) {
      if constexpr (HOST)
        __hip_atomic_store(acc_out + wave * O + lane, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        acc_out[wave * O + lane] = acc;
    }
    retire_touches(sink, c.norm_acc);
    collect_stamp(c, 1, 0);                          // staged + chained
    collect_signal_done(c, vb);
    collect_stamp(c, 1, 1);
    return;
  }
  // Actor: ONE 16-sample tile per workgroup; wave w owns output-feature tile w (16 of the 64
  // features) of both hidden layers, so the dependent chain per launch is 5 + 8 MFMAs and four
  // tanh per lane instead of 20 + 64 and 32.  h1 is exchanged through LDS (one b128 per lane),
  // the head is a per-wave partial dot product folded by wave 0.
  const PackedActor L(KS1, AP);
  const float* P = c.packed;
  const int lane = tid & 63, s = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: uniform branches
  const int64_t ntiles = (W + 15) / 16;
  f32x4* X1 = reinterpret_cast<f32x4*>(tile);                  // [4 tiles][64 lanes] h1 values
  float* ZP = tile + 1024;                                     // [4 waves][AP][16 samples]
  float eps_sink = 0.f;
  for (int64_t t = vb; t < ntiles; t += act_blocks) {
    const int64_t ns = t * 16 + s;
    const bool valid = ns < W;
    const int64_t nc = valid ? ns : W - 1;
    // Wave 0 finishes the step: lane (s, g) forms actions g and g + 4 of sample s (the four lane
    // groups used to repeat all AP of them: the one wave everybody waits for ran 6 dependent
    // tanh / log-density chains in a row where 2 do)
    constexpr int kMine = (AP + 3) / 4;
    int mine[kMine];
#pragma unroll
    for (int u = 0; u < kMine; ++u) mine[u] = min(g + 4 * u, AP - 1);
    float xr[KS1], ep[kMine];
    // HOST: the tile's observation and noise rows cross PCIe exactly once, as 16-byte requests
    // that are all in flight together; they meet in LDS (SO / SE) further down.
    const int tile_rows = (int)min<int64_t>(16, W - t * 16);
    const int64_t o_first = t * 16 * O, o_count = (int64_t)tile_rows * O, o_vecs = o_count >> 2;
    const int64_t e_first = t * 16 * A, e_count = (int64_t)tile_rows * A, e_vecs = e_count >> 2;
    const int etid = (tid + 128) & 255;                           // noise on the other two waves
    f32x4 vo = {0.f, 0.f, 0.f, 0.f}, ve = {0.f, 0.f, 0.f, 0.f};
    float to = 0.f, te = 0.f;
    // carry-over (Collect16Args::next_from_obs): the reset flags of the tile's workers ride along
    const bool carry = HOST && c.next_from_obs != 0 && c.outcome_row >= 0;        // scalar
    float reset_flag = 1.f;
    if constexpr (HOST) {
      if (carry && tid < tile_rows) reset_flag = host_load1<SYS>(c.resets, t * 16 + tid);
      if (tid < o_vecs)
        vo = host_load4<SYS>(c.obs + o_first, tid);
      if ((o_vecs << 2) + tid < o_count)
        to = host_load1<SYS>(c.obs + o_first, (o_vecs << 2) + tid);
      if (c.eps != nullptr) {
        if (etid < e_vecs)
          ve = host_load4<SYS>(c.eps + e_first, etid);
        if ((e_vecs << 2) + etid < e_count)
          te = host_load1<SYS>(c.eps + e_first, (e_vecs << 2) + etid);
      }
    } else {
#pragma unroll
      for (int st = 0; st < KS1; ++st) {
        const int k = 4 * st + g;
        xr[st] = c.obs[nc * O + (k < O ? k : O - 1)];
      }
      // (noise and head constants are wave 0's alone: every load instruction costs the CU's
      //  memory pipe 5 - 10 ns, and the four waves share it)
#pragma unroll
      for (int u = 0; u < kMine; ++u)
        ep[u] = (wave == 0 && c.eps != nullptr) ? c.eps[nc * A + min(mine[u], A - 1)] : 0.f;
    }
    // this tile's weight operands, all requested up front
    const f32x4 bias1 = reinterpret_cast<const f32x4*>(P + L.B1P + g * 16)[wave];
    const f32x4 bias2 = reinterpret_cast<const f32x4*>(P + L.B2P + g * 16)[wave];
    float wl[KS1];
    f32x4 w2[4], w3[AP], hcM[kMine];
#pragma unroll
    for (int st = 0; st < KS1; ++st) wl[st] = P[L.W1I + (wave * KS1 + st) * 64 + lane];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      w2[cc] = reinterpret_cast<const f32x4*>(P + L.W2S)[(wave * 4 + cc) * 64 + lane];
#pragma unroll
    for (int aa = 0; aa < AP; ++aa) {
      w3[aa] = reinterpret_cast<const f32x4*>(P + L.W3P)[(aa * 4 + g) * 4 + wave];
    }
#pragma unroll
    for (int u = 0; u < kMine; ++u)
      hcM[u] = wave == 0 ? *reinterpret_cast<const f32x4*>(P + L.HC + mine[u] * 8)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HOST) {
      float* SO = tile + 2048;                     // [16][O] observation rows of the tile
      float* SE = tile + 2048 + 512;               // [16][A] noise rows
      float* seg = c.seg_obs + c.row * W * O + o_first;
      if (tid < o_vecs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { SO[4 * tid + e] = vo[e]; seg[4 * tid + e] = vo[e]; }
      }
      if ((o_vecs << 2) + tid < o_count) {
        SO[(o_vecs << 2) + tid] = to;
        seg[(o_vecs << 2) + tid] = to;
      }
      if (c.eps != nullptr) {
        if (etid < e_vecs) {
#pragma unroll
          for (int e = 0; e < 4; ++e) SE[4 * etid + e] = ve[e];
        }
        if ((e_vecs << 2) + etid < e_count) SE[(e_vecs << 2) + etid] = te;
      }
      float* SR = tile + 2048 + 512 + 128;         // [16] reset flags of the tile's workers
      if (carry && tid < tile_rows) SR[tid] = reset_flag;
      __syncthreads();
      if (carry) {
        // the rows of the workers that did not reset ARE the previous step's next observations
        // (floor(at / O) = umulhi(at, ceil(2^32 / O)): at < 16 O)
        float* carried = c.seg_next + c.outcome_row * W * O + o_first;
        const unsigned row_magic = 0xffffffffu / (unsigned)O + 1u;
        auto row_of = [&](unsigned at) { return O == 1 ? at : __umulhi(at, row_magic); };
        if (tid < o_vecs) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned at = 4u * (unsigned)tid + e;
            if (SR[row_of(at)] == 0.f) carried[at] = vo[e];
          }
        }
        const unsigned at = (unsigned)(o_vecs << 2) + (unsigned)tid;
        if (at < (unsigned)o_count && SR[row_of(at)] == 0.f) carried[at] = to;
      }
      if (vb == 0) collect_stamp(c, 0, 0);   // inputs in LDS
      const int sr = s < tile_rows ? s : tile_rows - 1;
#pragma unroll
      for (int st = 0; st < KS1; ++st) {
        const int k = 4 * st + g;
        xr[st] = SO[sr * O + (k < O ? k : O - 1)];
      }
#pragma unroll
      for (int u = 0; u < kMine; ++u)
        ep[u] = (wave == 0 && c.eps != nullptr) ? SE[sr * A + min(mine[u], A - 1)] : 0.f;
    } else {
      // the tile's 16 observation rows -> Segment row (contiguous, coalesced)
      for (int64_t i = tid; i < o_count; i += 256)
        c.seg_obs[c.row * W * O + o_first + i] = c.obs[o_first + i];
      if (c.pf_eps != nullptr) {                  // next step's noise rows of this tile
        for (int64_t i = (int64_t)tid * 16; i < e_count; i += 256 * 16)
          touch(c.pf_eps + e_first + i, eps_sink);
      }
    }
    f32x4 acc = bias1;
#pragma unroll
    for (int st = 0; st < KS1; ++st) {
      const float xv = xr[st] * ((valid && 4 * st + g < O) ? 1.f : 0.f);
      acc = mfma16(wl[st], xv, acc);
    }
    f32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = tanh_fast(acc[e]);
    X1[wave * 64 + lane] = h;
    __syncthreads();
    if (vb == 0) collect_stamp(c, 0, 1);     // layer 1 done (inputs arrived before)
    f32x4 h1[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) h1[cc] = X1[cc * 64 + lane];
    f32x4 even = bias2, odd = {0.f, 0.f, 0.f, 0.f};            // two chains: half the latency
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e & 1) odd = mfma16(w2[cc][e], h1[cc][e], odd);
        else even = mfma16(w2[cc][e], h1[cc][e], even);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = tanh_fast(even[e] + odd[e]);
    if (vb == 0) collect_stamp(c, 0, 2);     // layer 2 done
#pragma unroll
    for (int aa = 0; aa < AP; ++aa) {
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(h[e], w3[aa][e], part);
      part = sum_groups(part);
      if (g == 0) ZP[(wave * AP + aa) * 16 + s] = part;
    }
    __syncthreads();
    if (vb == 0) collect_stamp(c, 0, 3);     // head partials exchanged
    if (wave == 0) {
      // Three straight-line stages — all LDS reads, then the AP independent head chains, then the
      // stores under ONE branch.  (Stores inside the per-action loop made every action its own
      // exec-masked block: read -> wait -> tanh chain -> store, six times in a row, 1.4 us of
      // the step's 5.8 us on the one wave everybody waits for.)
      float zp[kMine][4], actv[kMine], term[kMine];
#pragma unroll
      for (int u = 0; u < kMine; ++u) {
#pragma unroll
        for (int w = 0; w < 4; ++w) zp[u][w] = ZP[(w * AP + mine[u]) * 16 + s];
      }
#pragma unroll
      for (int u = 0; u < kMine; ++u) {
        const float z = (zp[u][0] + zp[u][1]) + (zp[u][2] + zp[u][3]);
        const f32x4 hc = hcM[u];
        const float loc = tanh_fast(z + hc[0]);
        actv[u] = c.eps != nullptr ? loc + hc[1] * ep[u] : loc;
        const float d = actv[u] - loc;
        term[u] = g + 4 * u < A ? -(d * d) * hc[2] - hc[3] : 0.f;
      }
      // the log-probability: the terms of all actions, added in action order like before
      float logp = 0.f;
#pragma unroll
      for (int aa = 0; aa < AP; ++aa) logp += __shfl(term[aa >> 2], s + 16 * (aa & 3), 64);
      if (valid) {
#pragma unroll
        for (int u = 0; u < kMine; ++u) {
          const int aa = g + 4 * u;
          if (aa < A) {
            c.seg_act[(c.row * W + ns) * A + aa] = actv[u];
            if (c.actions_out != nullptr) {
#if TONIC_COLLECT_SC1
              if constexpr (HOST)
                __hip_atomic_store(c.actions_out + ns * A + aa, actv[u], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
              else
#endif
                c.actions_out[ns * A + aa] = actv[u];
            }
          }
        }
        if (g == 0) c.seg_lp[c.row * W + ns] = logp;
      }
      // HOST: the actions sit in this XCD's L2 until a system-scope release writes them back;
      // only then may the completion word go out (scripts/collector_stress.py: without the
      // fence the host reads stale actions within a few thousand steps).
      if (vb == 0) collect_stamp(c, 0, 4);   // head, sample, stores issued
#if !TONIC_COLLECT_SC1
      if constexpr (HOST) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
      if (vb == 0) collect_stamp(c, 0, 5);   // released
    }
    __syncthreads();                                 // X1 / ZP are reused by the next tile
  }
  retire_touches(eps_sink, c.seg_lp);
  collect_signal_done(c, vb);
  collect_signal_rows(c, vb);
  if (vb == 0) collect_stamp(c, 0, 6);       // flag out
}

// One launch per environment step.
template <int KS1, int AP, bool HOST>
__global__ __launch_bounds__(256) void ppo_collect16_kernel(Collect16Args c) {
  __shared__ __attribute__((aligned(16))) float tile[kCollectLds];
  collect16_step<KS1, AP, HOST, false>(c, tile, (int)blockIdx.x, (int)gridDim.x);
}

// RESIDENT form for the pinned-host collector (transport 2): launched once per rollout, every
// workgroup keeps its slot and waits for the host's next command word instead of being launched
// again — the launch (3.4 us of host call + ~4 us until the first wave runs) and the cold start
// of a fresh kernel leave the critical path of an environment step (17.7 -> ~9 us from command to
// actions; scripts/ubench/pingpong.hip measures 5.4 us for the bare exchange of these bytes).
//   command word (pinned host memory, written by the host with one 8-byte store):
//     [63:32] sequence number | [31:8] Segment row | bit 3 stop | bit 2 store the previous
//     outcome | bit 1 noise on | bit 0 noise slot
// Thread 0 of every workgroup polls it with system-scope loads.
//
// Who runs a slot (round 4).  The grid has one workgroup per slot, but a slot's work does not wait
// for ITS workgroup: on a GPU that is busy with other kernels some workgroups of the launch get a
// compute unit late, or only when that other work ends.  Every (slot, command) is run exactly once
// by whoever holds its claim — `claims[slot]` (device memory, zero at launch) is the newest command
// claimed for the slot, moved with atomic max:
//   * a workgroup that is there claims its own slot one command AHEAD, with a no-return atomic
//     issued before it runs the current command — the host cannot issue the next command before
//     this slot's completion word, which goes out behind that atomic — so the steady state has no
//     atomic round trip on the step's critical path;
//   * having run its own slot, a workgroup looks at the slots right before its own (cyclically):
//     those whose claim is older than the command belong to workgroups that are not there; it
//     claims and runs them in slot order (actor tiles before the copy and record slots, which wait
//     for them) up to the first slot that is claimed.  One load of a neighbour's claim per command
//     when everybody is there, after the completion word is out;
//   * a workgroup that arrives late claims with a RETURNING atomic and skips what was already run
//     for it (the host may have moved on: running a finished command again would read the block's
//     next inputs).
// Nobody waits forever: `park_ticks` of the 100 MHz wall clock after it has finished a command's LAST
// slot (the record: it waits for the actor slots' rows, so the whole step is done by then) without a
// new command, the workgroup that ran that slot announces that the kernel parks — to the others
// through `relay` (device memory), to the host through `parked` (the command it was waiting for) —
// and leaves; so does everybody who sees the notice (and, after 1 000 x as long, everybody anyway).
// The host launches the kernel again (claims zeroed) when it has the next command; a command that
// arrived during the announcement and was run by some is run again by the new launch (a step is
// idempotent while the host still waits for it, and the host waits for every slot of the new launch).
template <int KS1, int AP>
__global__ __launch_bounds__(256) void ppo_collect_resident_kernel(Collect16Args c,
                                                                   CollectResident r) {
  __shared__ __attribute__((aligned(16))) float tile[kCollectLds];
  __shared__ unsigned long long command;
  __shared__ int verdict[2];                       // {run this slot?, slots to help with} from thread 0
  const int nb = (int)gridDim.x, me = (int)blockIdx.x;
  const int act_blocks = nb - 1 - kCollectCopyBlocks;
  unsigned last = r.first_seq - 1u;                // the newest command this workgroup has dealt with
  unsigned held = last;                            // ... and the newest one its own slot is claimed for
  // whoever ran the last slot of the newest command keeps the clock of the idle period behind it (at
  // launch: that slot's own workgroup)
  bool announcer = me == nb - 1;
  for (;;) {
    if (threadIdx.x == 0) {
      unsigned long long word;
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        word = __hip_atomic_load(r.command, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)((unsigned)(word >> 32) - last) > 0) break;       // a command not dealt with yet
        const unsigned long long waited = wall_clock64() - t0;
        if (waited > (announcer ? r.park_ticks : 1000ull * r.park_ticks)) {
          __hip_atomic_store(r.relay, last + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(r.parked, last + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (c.stamps != nullptr) __hip_atomic_fetch_add(c.stamps + 26, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          word = 0;                                // leave (no command has sequence number 0)
          break;
        }
        // the others' notice is looked at only after 20 us without a command: a load that is
        // waited for delays the next poll
        if (waited > 2000 &&
            __hip_atomic_load(r.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          word = 0;
          break;
        }
        for (int i = 0; i < r.poll_sleep; ++i) __builtin_amdgcn_s_sleep(8);      // (~0.25 us each)
      }
      if (word != 0) {
        // my own slot: claimed ahead (steady state: nothing to wait for), or asked for now
        const unsigned seq = (unsigned)(word >> 32);
        if ((int)(held - seq) >= 0) {
          verdict[0] = 1;
          (void)__hip_atomic_fetch_max(r.claims + me, seq + 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
          held = seq + 1u;
        } else {
          const unsigned old = __hip_atomic_fetch_max(r.claims + me, seq + 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
          verdict[0] = (int)(old - seq) < 0 ? 1 : 0;     // older than this command: nobody ran it for me
          if (c.stamps != nullptr && verdict[0] == 0)
            __hip_atomic_fetch_add(c.stamps + 25, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (old == seq: somebody ran this one for me, the next is mine; old > seq: the host has
          //  moved on and somebody holds that one too — ask again at the next command)
          held = (int)(old - seq) > 0 ? seq - 1u : seq + 1u;
        }
      }
      command = word;
    }
    __syncthreads();
    const unsigned long long word = command;
    if (word == 0) return;                         // parked
    const unsigned seq = (unsigned)(word >> 32);
    Collect16Args step = c;
    step.abandon = r.relay;
    step.row = (int64_t)((word >> 8) & 0xffffff);
    step.outcome_row = (word & 4) ? step.row - 1 : -1;
    step.eps = (word & 2) ? ((word & 1) ? r.eps1 : r.eps0) : nullptr;
    step.done_seq = seq;
    step.stamp_t0 = wall_clock64();
    const bool stop = (word & 8) != 0;             // stop: only the pending outcome is stored
    if (stop) step.next_from_obs = 0;              // (nobody acts: no observation rows to carry over)
    // this command: my own slot first, then the slots of workgroups that are not there
    int vb = me, helping = -1, next = 0;
    bool go = verdict[0] != 0;
    announcer = false;
    for (;;) {
      if (go) {
        const bool copy_slot = vb >= act_blocks && vb < nb - 1;
        announcer = announcer || vb == nb - 1;
        if (stop && !copy_slot) collect_signal_done(step, vb);
        else collect16_step<KS1, AP, true, true>(step, tile, vb, nb);
      }
      __syncthreads();                             // (`tile`, `verdict` are reused)
      if (helping < 0) {
        if (threadIdx.x == 0) {
          int count = 0, b = me;
          while (count < nb - 1) {
            b = b == 0 ? nb - 1 : b - 1;
            const unsigned theirs = __hip_atomic_load(r.claims + b, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(theirs - seq) >= 0) break;   // claimed: its holder also looks after the ones before it
            ++count;
          }
          verdict[1] = count;
        }
        __syncthreads();
        helping = verdict[1];
        next = me - helping;                       // the run [me - helping, me), in slot order
        if (next < 0) next += nb;
      }
      if (helping == 0) break;
      vb = next;
      next = next + 1 == nb ? 0 : next + 1;
      --helping;
      if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_max(r.claims + vb, seq, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        verdict[0] = (int)(old - seq) < 0 ? 1 : 0;
        if (c.stamps != nullptr && verdict[0] != 0)
          __hip_atomic_fetch_add(c.stamps + 24, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      go = verdict[0] != 0;
    }
    last = seq;
    if (stop) return;
  }
}

// ------------------------------------------------------------------------------- host side

bool grad16_supported(int O, int A, bool actor) {
  return O >= 1 && O <= 32 && (!actor || (A >= 1 && A <= 8));
}

int grad16_blocks(int64_t n) {
  const int64_t tiles = (n + 15) / 16;
  int64_t blocks = (tiles + kWaves16 - 1) / kWaves16;
  if (blocks > 256) blocks = 256;
  return (int)(blocks < 1 ? 1 : blocks);
}

namespace {

template <int KS1, int XT, int XR, int AP, bool ACTOR, bool EXACT, int CH>
int go16(int blocks, hipStream_t stream, const MlpArgs& args) {
  auto kernel = mlp64_grad16_kernel<KS1, XT, XR, AP, ACTOR, EXACT, CH>;
  constexpr int lds_bytes = Lds16<KS1, AP, CH>::BYTES;
  static_assert(lds_bytes <= 160 * 1024, "mlp64_grad16: LDS budget");
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("mlp64_grad16: hipFuncSetAttribute(%d B LDS): %s", lds_bytes, hipGetErrorString(e));
      return TONIC_ERR_LAUNCH;
    }
    configured = true;
  }
  // the epilogue overlays five gradient images on the weight / tile areas
  const int P = ACTOR ? 64 * args.O + 64 + 4096 + 64 + args.A + 64 * args.A + args.A
                      : 64 * args.O + 64 + 4096 + 64 + 64 + 1;
  TONIC_REQUIRE(5 * ((P + kStatSlots + 63) / 64 * 64) * 4 <= lds_bytes, TONIC_ERR_INVALID_ARGUMENT,
                "mlp64_grad16: gradient images do not fit the %d B of LDS", lds_bytes);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kWaves16 * 64), lds_bytes, stream, args);
  TONIC_CHECK_LAUNCH("mlp64_grad16_kernel");
  return TONIC_OK;
}

template <int KS1, int XT, int XR, int CH>
int go16_values(int blocks, hipStream_t stream, const MlpArgs& args) {
  auto kernel = mlp64_grad16_kernel<KS1, XT, XR, 1, false, true, CH, false, true>;
  constexpr int lds_bytes = Lds16<KS1, 1, CH>::BYTES;
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      set_error("mlp64_grad16<values>: hipFuncSetAttribute(%d B LDS): %s", lds_bytes,
                hipGetErrorString(e));
      return TONIC_ERR_LAUNCH;
    }
    configured = true;
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kWaves16 * 64), lds_bytes, stream, args);
  TONIC_CHECK_LAUNCH("mlp64_grad16_kernel<values>");
  return TONIC_OK;
}

template <int CH>
int values_by_inputs(int blocks, hipStream_t stream, const MlpArgs& args) {
  if (args.O <= 4) return go16_values<1, 0, 4, CH>(blocks, stream, args);
  if (args.O <= 16) return go16_values<4, 1, 0, CH>(blocks, stream, args);
  if (args.O == 17) return go16_values<5, 1, 1, CH>(blocks, stream, args);
  if (args.O <= 20) return go16_values<5, 1, 4, CH>(blocks, stream, args);
  return go16_values<8, 2, 0, CH>(blocks, stream, args);
}

template <int KS1, int XT, int XR, int CH>
int by_heads(bool actor, int blocks, hipStream_t st, const MlpArgs& a) {
  if (!actor) return go16<KS1, XT, XR, 1, false, true, CH>(blocks, st, a);
  if (a.A == 1) return go16<KS1, XT, XR, 1, true, true, CH>(blocks, st, a);
  if (a.A == 6) return go16<KS1, XT, XR, 6, true, true, CH>(blocks, st, a);
  if (a.A == 8) return go16<KS1, XT, XR, 8, true, true, CH>(blocks, st, a);
  if (a.A < 6) return go16<KS1, XT, XR, 6, true, false, CH>(blocks, st, a);
  return go16<KS1, XT, XR, 8, true, false, CH>(blocks, st, a);
}

template <int CH>
int by_inputs(bool actor, int blocks, hipStream_t stream, const MlpArgs& args) {
  // (x steps KS1 = ceil(O/4), full 16-column dW1 tiles, remainder columns)
  if (args.O <= 4) return by_heads<1, 0, 4, CH>(actor, blocks, stream, args);
  if (args.O <= 16) return by_heads<4, 1, 0, CH>(actor, blocks, stream, args);
  if (args.O == 17) return by_heads<5, 1, 1, CH>(actor, blocks, stream, args);     // HalfCheetah
  if (args.O <= 20) return by_heads<5, 1, 4, CH>(actor, blocks, stream, args);
  return by_heads<8, 2, 0, CH>(actor, blocks, stream, args);
}

}  // namespace

int launch_grad16_probe(int blocks, hipStream_t stream, const MlpArgs& args) {
  auto kernel = mlp64_grad16_kernel<5, 1, 1, 6, true, true, 3, true>;     // the shipped arithmetic
  constexpr int lds_bytes = Lds16<5, 6, 3>::BYTES;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) { set_error("probe: %s", hipGetErrorString(e)); return TONIC_ERR_LAUNCH; }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kWaves16 * 64), lds_bytes, stream, args);
  TONIC_CHECK_LAUNCH("mlp64_grad16_kernel<probe>");
  return TONIC_OK;
}

// The critic's forward over a whole batch: values -> args.out1 (chain as launch_grad16)
// chain 3 (fp16x2 terms) is the ONE form the product library holds; 0 - 2 (fp32 MFMA, bf16x3) are the
// references the parity tests compare it with, built only with TONIC_DEV (libtonic_hip_dev.so).
int launch_values16(int blocks, hipStream_t stream, const MlpArgs& args, int chain) {
  if (chain == 3) return values_by_inputs<3>(blocks, stream, args);
#ifdef TONIC_DEV
  if (chain == 2) return values_by_inputs<2>(blocks, stream, args);
  return chain == 1 ? values_by_inputs<1>(blocks, stream, args)
                    : values_by_inputs<0>(blocks, stream, args);
#else
  set_error("grad chain %d is a developer reference: build with TONIC_DEV=1", chain);
  return TONIC_ERR_INVALID_ARGUMENT;
#endif
}

int launch_grad16(bool actor, int blocks, hipStream_t stream, const MlpArgs& args, int chain) {
  if (chain == 3) return by_inputs<3>(actor, blocks, stream, args);
#ifdef TONIC_DEV
  if (chain == 2) return by_inputs<2>(actor, blocks, stream, args);
  return chain == 1 ? by_inputs<1>(actor, blocks, stream, args)
                    : by_inputs<0>(actor, blocks, stream, args);
#else
  set_error("grad chain %d is a developer reference: build with TONIC_DEV=1", chain);
  return TONIC_ERR_INVALID_ARGUMENT;
#endif
}

}  // namespace tonic

using namespace tonic;

int tonic::collect16_ks1(int O) { return O <= 4 ? 1 : O <= 20 ? 5 : 8; }
int tonic::collect16_ap(int A) { return A <= 1 ? 1 : A <= 6 ? 6 : 8; }

int tonic::launch_actor_pack(const float* d_actor_params, float* d_packed, int O, int A,
                             hipStream_t stream) {
  hipLaunchKernelGGL(actor_pack_kernel, dim3(8), dim3(256), 0, stream, d_actor_params, d_packed,
                     O, A, collect16_ks1(O), collect16_ap(A));
  TONIC_CHECK_LAUNCH("tonic_ppo_pack_actor");
  return TONIC_OK;
}

extern "C" int64_t tonic_ppo_packed_actor_floats(int32_t O, int32_t A) {
  return PackedActor(collect16_ks1(O), collect16_ap(A)).total;
}

extern "C" int tonic_ppo_pack_actor(const float* d_actor_params, float* d_packed, int32_t O,
                                    int32_t A, void* stream) {
  TONIC_REQUIRE(d_actor_params && d_packed && O >= 1 && O <= 32 && A >= 1 && A <= 8,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_pack_actor: bad argument");
  return launch_actor_pack(d_actor_params, d_packed, O, A, as_stream(stream));
}

int tonic::collect16_blocks(int64_t W) {
  const int64_t tiles = (W + 15) / 16;
  return (int)(tiles < 4096 ? tiles : 4096) + kCollectCopyBlocks + 1;   // one tile per workgroup
}

int tonic::launch_collect_resident(const Collect16Args& c, const CollectResident& r,
                                   hipStream_t st) {
  const dim3 grid(collect16_blocks(c.W)), block(256);
  const int ks1 = collect16_ks1(c.O), ap = collect16_ap(c.A);
#define TONIC_RESIDENT(K, P_)                                                              \
  if (ks1 == K && ap == P_) {                                                              \
    hipLaunchKernelGGL((ppo_collect_resident_kernel<K, P_>), grid, block, 0, st, c, r);    \
  } else
  TONIC_RESIDENT(1, 1) TONIC_RESIDENT(1, 6) TONIC_RESIDENT(1, 8)
  TONIC_RESIDENT(5, 1) TONIC_RESIDENT(5, 6) TONIC_RESIDENT(5, 8)
  TONIC_RESIDENT(8, 1) TONIC_RESIDENT(8, 6) TONIC_RESIDENT(8, 8) {}
#undef TONIC_RESIDENT
  TONIC_CHECK_LAUNCH("ppo_collect_resident_kernel");
  return TONIC_OK;
}

int tonic::launch_collect16(const Collect16Args& c, hipStream_t st) {
  const dim3 grid(collect16_blocks(c.W)), block(256);
  const int ks1 = collect16_ks1(c.O), ap = collect16_ap(c.A);
  const bool host = c.done_flags != nullptr;          // the pinned-host collector, transport 0
#define TONIC_COLLECT16(K, P_)                                                        \
  if (ks1 == K && ap == P_) {                                                         \
    if (host) hipLaunchKernelGGL((ppo_collect16_kernel<K, P_, true>), grid, block, 0, st, c);   \
    else hipLaunchKernelGGL((ppo_collect16_kernel<K, P_, false>), grid, block, 0, st, c);       \
  } else
  TONIC_COLLECT16(1, 1) TONIC_COLLECT16(1, 6) TONIC_COLLECT16(1, 8)
  TONIC_COLLECT16(5, 1) TONIC_COLLECT16(5, 6) TONIC_COLLECT16(5, 8)
  TONIC_COLLECT16(8, 1) TONIC_COLLECT16(8, 6) TONIC_COLLECT16(8, 8) {}
#undef TONIC_COLLECT16
  TONIC_CHECK_LAUNCH("ppo_collect16_kernel");
  return TONIC_OK;
}

extern "C" int tonic_ppo_collect_steps_packed(
    const float* d_packed_actor, const float* d_observations, const float* d_eps,
    const float* d_rewards, const float* d_resets, const float* d_terminations,
    float* d_seg_observations, float* d_seg_actions, float* d_seg_next_observations,
    float* d_seg_rewards, float* d_seg_resets, float* d_seg_terminations,
    float* d_seg_log_probs, float* d_norm_acc, int64_t row0, int64_t steps, int64_t W,
    int32_t O, int32_t A, void* stream) {
  TONIC_REQUIRE(d_packed_actor && d_observations && d_rewards && d_resets && d_terminations &&
                    d_seg_observations && d_seg_actions && d_seg_next_observations &&
                    d_seg_rewards && d_seg_resets && d_seg_terminations && d_seg_log_probs &&
                    row0 >= 0 && steps > 0 && W > 0 && O >= 1 && O <= 32 && A >= 1 && A <= 8,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_collect_steps_packed: bad argument");
  hipStream_t st = as_stream(stream);
  for (int64_t t = 0; t < steps; ++t) {
    const float* obs = d_observations + t * W * O;
    Collect16Args c{d_packed_actor, obs, d_eps ? d_eps + t * W * A : nullptr, obs + W * O,
                    d_rewards + t * W, d_resets + t * W, d_terminations + t * W,
                    d_seg_observations, d_seg_actions, d_seg_next_observations, d_seg_rewards,
                    d_seg_resets, d_seg_terminations, d_seg_log_probs, d_norm_acc, nullptr,
                    row0 + t, W, 0, O, A, nullptr, nullptr, nullptr, nullptr, nullptr,
                    row0 + t, nullptr, 0u};
    if (t + 1 < steps) {                         // the next step's inputs exist: touch them early
      c.pf_eps = d_eps ? d_eps + (t + 1) * W * A : nullptr;
      c.pf_next_obs = obs + 2 * W * O;
      c.pf_rewards = d_rewards + (t + 1) * W;
      c.pf_resets = d_resets + (t + 1) * W;
      c.pf_terminations = d_terminations + (t + 1) * W;
    }
    launch_collect16(c, st);
    TONIC_CHECK_LAUNCH("tonic_ppo_collect_steps_packed");
  }
  return TONIC_OK;
}

extern "C" int tonic_ppo_collect_step_packed(
    const float* d_packed_actor, const float* d_observations, const float* d_eps,
    const float* d_next_observations, const float* d_rewards, const float* d_resets,
    const float* d_terminations, float* d_seg_observations, float* d_seg_actions,
    float* d_seg_next_observations, float* d_seg_rewards, float* d_seg_resets,
    float* d_seg_terminations, float* d_seg_log_probs, float* d_norm_acc, float* d_actions_out,
    int64_t row, int64_t W, int32_t O, int32_t A, void* stream) {
  TONIC_REQUIRE(d_packed_actor && d_observations && d_next_observations && d_rewards &&
                    d_resets && d_terminations && d_seg_observations && d_seg_actions &&
                    d_seg_next_observations && d_seg_rewards && d_seg_resets &&
                    d_seg_terminations && d_seg_log_probs && row >= 0 && W > 0 && O >= 1 &&
                    O <= 32 && A >= 1 && A <= 8,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_ppo_collect_step_packed: bad argument");
  Collect16Args c{d_packed_actor, d_observations, d_eps, d_next_observations, d_rewards,
                  d_resets, d_terminations, d_seg_observations, d_seg_actions,
                  d_seg_next_observations, d_seg_rewards, d_seg_resets, d_seg_terminations,
                  d_seg_log_probs, d_norm_acc, d_actions_out, row, W, 0, O, A,
                  nullptr, nullptr, nullptr, nullptr, nullptr, row, nullptr, 0u};
  launch_collect16(c, as_stream(stream));
  TONIC_CHECK_LAUNCH("tonic_ppo_collect_step_packed");
  return TONIC_OK;
}
