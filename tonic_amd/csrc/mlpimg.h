// fp16x2 operand-order WEIGHT IMAGES of the off-policy networks (round 6).
//
// Why: in the fused passes of mlpfwd.hip one workgroup of four waves carries 16 batch rows through a whole
// network, so every layer-sized phase has to (a) stream a whole weight matrix from L2 and (b) issue its products.
// On fp32 MFMAs over the padded row-major parameter block both cost ~4 us per 256 x 256 layer and neither can be
// dropped alone (profiles/r03_offpolicy_phases.md); measured in the layer loop's own harness
// (profiles/r05_ubench_layer_f16x2.md): the same layer as THREE v_mfma_f32_16x16x32_f16 per product on two-term
// binary16 splits (hi + lo: 22+ significant bits; the dropped lo.lo is <= 2^-22 |a||b|), its weights read from an
// image that is already in MFMA A-operand order (one contiguous 1 KB block per tile, k-chunk and term: the
// stream runs at ~120 B/ns per CU instead of 62 - 70), takes 2.1 us, 2.8 us with the whole activation epilogue.
//
// An image of a matrix A [M][K] (A operand: M = the product's output features, K = the contraction):
//   [tiles = ceil(M / 16)][chunks = ceil(K / 32)][2 terms: hi, lo][64 lanes][8 binary16]
//   lane (m = lane & 15, g = lane >> 4) of block (t, c) holds A[16 t + m][32 c + 8 g + e], e = 0 .. 7, times 2^7
//   (kImgScaleExp: |w| < 512 representable, absolute resolution 2^-32; a weight beyond that becomes inf in the
//   image and NaN in every product — loud, like a diverged run of the reference); rows >= M and columns >= K
//   hold zeros and stay zero.
// Per network: the forward images of W1, W2 and the policy heads, and the transposed ones the input-gradient
// chains walk (W2^T; the heads' ^T; the action columns of a critic's W1^T).  They live in the caller's workspace,
// NOT in the parameter blocks: `build_weight_images` forms them from the float32 parameters (the authority) at
// the start of every update call / split entry point, and inside a fused update the optimizer epilogue of the
// weight-gradient launch (gemm_tn_tile, gemm16.hip) writes the new parameter's — and its polyak target's — image
// entries next to the float32 values, with the same conversion (img_terms), so a rebuilt image and an
// epilogue-maintained one hold the same bits.
#pragma once
#include "common.h"

namespace tonic {

constexpr int kImgScaleExp = 7;
constexpr int kImgTermBytes = 1024;              // one term of one (tile, chunk) block: 64 lanes x 16 bytes

struct ImgView {
  int64_t off;                                   // bytes from the start of the network's image block
  int tiles, chunks;
  __host__ __device__ int64_t bytes() const { return (int64_t)tiles * chunks * 2 * kImgTermBytes; }
};

// Index (in binary16 units, term 0) of element (r, k) inside an image with `chunks` k-chunks; the lo term
// follows kImgTermBytes / 2 halfs later.
__host__ __device__ inline int64_t img_half_index(int chunks, int r, int k) {
  const int t = r >> 4, m = r & 15, c = k >> 5, g = (k >> 3) & 3, e = k & 7;
  return ((((int64_t)t * chunks + c) * 2) * 64 + g * 16 + m) * 8 + e;
}

struct ActorImages { ImgView f1, f2, fh[2], t2, th[2]; int64_t bytes; };
struct CriticImages { ImgView f1, f2, t2, t1a; int64_t bytes; };

__host__ __device__ inline ImgView img_take(int64_t& at, int M, int K) {
  ImgView v{at, (M + 15) / 16, (K + 31) / 32};
  at += v.bytes();
  return v;
}
// actor: W1 [H, O], W2 [H, H], heads [A, H] (x heads)
__host__ __device__ inline ActorImages actor_images(int O, int H, int A, int heads) {
  ActorImages im{};
  int64_t at = 0;
  im.f1 = img_take(at, H, O);
  im.f2 = img_take(at, H, H);
  im.t2 = img_take(at, H, H);
  for (int h = 0; h < 2; ++h) {
    im.fh[h] = h < heads ? img_take(at, A, H) : im.fh[0];
    im.th[h] = h < heads ? img_take(at, H, A) : im.th[0];
  }
  im.bytes = at;
  return im;
}
// critic: W1 [H, O + A], W2 [H, H]; t1a = the action columns of W1 transposed ([A][H])
__host__ __device__ inline CriticImages critic_images(int O, int A, int H) {
  CriticImages im{};
  int64_t at = 0;
  im.f1 = img_take(at, H, O + A);
  im.f2 = img_take(at, H, H);
  im.t2 = img_take(at, H, H);
  im.t1a = img_take(at, A, H);
  im.bytes = at;
  return im;
}

// Where the optimizer epilogue of a weight-gradient problem (one weight tensor W [M rows][N cols]) keeps the
// tensor's images up to date.  Pointers are the images of network 0 of the launch; `stride` bytes on per network
// (blockIdx.z); `target_delta`: the polyak target's image block minus the online one's.
struct ImgTarget {
  char* fwd; int fwd_chunks;                     // A[r][k] = W[r][k]                      (null: none)
  char* bwd; int bwd_chunks;                     // A'[k - col0][r] = W[r][k], col0 <= k < col0 + cols
  int bwd_col0, bwd_cols;
  int64_t stride, target_delta;
};

#if defined(__HIPCC__)
typedef unsigned img_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned img_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 img_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 img_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned img_pack(float a, float b) {       // v_cvt_pk_f16_f32, round to nearest even
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, img_f16x2));
}
// (a, b) -> packed hi and lo terms: a = hi + lo + r, |r| <= 2^-23 |a| (2^-25 absolute on the subnormal grid);
// the residual a - hi is exact in fp32: one v_fma_mix_f32 (binary16 half of `hi` x -1 + a)
__device__ __forceinline__ void img_split(float a, float b, unsigned& hi, unsigned& lo) {
  hi = img_pack(a, b);
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(b) : "v"(hi), "v"(b));
  lo = img_pack(a, b);
}
__device__ __forceinline__ float img_pow2(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }
// the image's terms of two weights (neighbours along K)
__device__ __forceinline__ void img_terms(float w0, float w1, unsigned& hi, unsigned& lo) {
  img_split(w0 * img_pow2(kImgScaleExp), w1 * img_pow2(kImgScaleExp), hi, lo);
}
// elements (r, k) and (r, k + 1), k even, of an image
__device__ __forceinline__ void img_store_pair(char* image, int chunks, int r, int k, float w0, float w1) {
  unsigned hi, lo;
  img_terms(w0, w1, hi, lo);
  char* at = image + 2 * img_half_index(chunks, r, k);
  *reinterpret_cast<unsigned*>(at) = hi;
  *reinterpret_cast<unsigned*>(at + kImgTermBytes) = lo;
}
#endif

// One conversion of build_weight_images: source tensor W [rows][cols] (rows `ld` floats apart) -> image.
//   plain:      A[r][k]        = W[r][k]                       (M = rows, K = cols)
//   transposed: A[k - col0][r] = W[r][k], col0 <= k < col0 + ncols   (M = ncols, K = rows)
struct ImgJob {
  const float* src; int ld, rows, cols;
  int transposed, col0, ncols;
  char* dst; int tiles, chunks;
};
constexpr int kImgJobsMax = 44;
struct ImgJobs { ImgJob job[kImgJobsMax]; int first[kImgJobsMax + 1]; int count; };

// host-side list of conversions (offpolicy.hip fills it per entry point)
struct ImgBuild {
  ImgJobs jobs{};
  void add(const float* src, int ld, int rows, int cols, char* block, ImgView v, bool transposed = false,
           int col0 = 0, int ncols = 0);
};
int launch_build_images(const ImgBuild& b, hipStream_t stream);

// The weights of one network pass as images (MlpFwdArgs / MlpBwdArgs): null `block` = the float32 passes.
struct FwdImages { const char* block; ImgView f1, f2, fh[2]; int64_t stride, second; };   // (stride per network; second: like second_params)
struct BwdImages { const char* block; ImgView t2, th[2], t1a; int64_t stride; };

}  // namespace tonic
