// The passes of mlpfwd.hip on fp16x2 terms from weight images (mlpimg.h) — included by mlpfwd.hip inside its
// anonymous namespace.  Same decomposition as the float32 passes (one workgroup of four waves = 16 batch rows
// through a whole network, wave w owns the 16-feature tiles w, w + 4, w + 8, w + 12, products formed as
// D[feature][row], activations exchanged through LDS), but
//   * a product is THREE v_mfma_f32_16x16x32_f16 (lo.hi, hi.lo, hi.hi) on two-term binary16 splits, fp32
//     accumulation: 96 MFMAs of 16 cycles per wave and 256 x 256 layer instead of 256 of 32;
//   * the A operand (weights) comes from an operand-order image: one buffer_load_b128 per tile, k-chunk (32) and
//     term, contiguous 1 KB blocks (~120 B/ns per CU from L2 against 62 - 70 for the padded rows);
//   * the B operand (the 16 rows' activations / gradients) lives in LDS as binary16 hi / lo rows in the unit of
//     each ROW's largest entry (ReLU activations and gradients have no bound: 2^(14 - ex), exact), written by the
//     previous layer's epilogue: bias + ReLU (or the ReLU mask), fp32 row to HBM for the weight gradients, row
//     maximum (lane groups by shuffles, waves through LDS), split, image.
// Accuracy: every operand keeps 22+ significant bits, the dropped lo.lo term is <= 2^-22 |a||b|: float32 class
// (the parity tests hold the passes to the torch-CPU oracle at 1e-5 like before).

__device__ __forceinline__ f32x4 mfma_h16(const img_u32x4& a, const img_u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(img_f16x8, a), __builtin_bit_cast(img_f16x8, b),
                                                c, 0, 0, 0);
}

constexpr int kActTop = 14;                      // a row's largest entry enters a product below 2^14
constexpr int kImgMaxChunks = 8;                 // H <= 256: eight k-chunks of 32

__host__ __device__ inline int pad32(int k) { return (k + 31) & ~31; }

// LDS of the image passes (bytes).  Region A: [2 terms][16 rows][pa halfs] — the first layer's input, later the
// second hidden layer's activations (forward) / dz2 (backward); region B: h1 / dz1; F32: the policy tail's head
// outputs (forward) / the formed head gradients (backward); MISC: row maxima, value partials, descale factors.
struct ImgLds {
  int pa, pb, off_a, off_b, off_misc, off_f32, total;
};
__host__ __device__ inline ImgLds img_lds(int K1, int H) {
  ImgLds L{};
  const int ka = pad32(K1 > H ? K1 : H);
  L.pa = ka + 8; L.pb = pad32(H) + 8;
  L.off_a = 0;
  L.off_b = L.off_a + 2 * kRows * L.pa * 2;
  L.off_misc = L.off_b + 2 * kRows * L.pb * 2;
  L.off_f32 = L.off_misc + (64 + 64 + 32) * 4;
  const int f32_floats = 3 * kRows * kPostPitch > 2 * kRows * kHeadPitch ? 3 * kRows * kPostPitch
                                                                       : 2 * kRows * kHeadPitch;
  L.total = L.off_f32 + f32_floats * 4;
  return L;
}
constexpr int kImgLdsCap = 128 * 1024;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_buffer(const char* image) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ img_u32x4 image16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned block_bytes) {
  return __builtin_bit_cast(img_u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, block_bytes, 0));
}

// max over the 16 lanes of a DPP row (= the 16 threads that share one batch row in the (row, slot) mappings)
template <int CTRL>
__device__ __forceinline__ float img_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float img_row_max16(float v) {
  v = fmaxf(v, img_dpp<0xB1>(v));      // quad_perm [1, 0, 3, 2]
  v = fmaxf(v, img_dpp<0x4E>(v));      // quad_perm [2, 3, 0, 1]
  v = fmaxf(v, img_dpp<0x141>(v));     // row_half_mirror
  v = fmaxf(v, img_dpp<0x140>(v));     // row_mirror
  return v;
}
// the exponent ex of a row maximum (mx < 2^ex), clamped so that both 2^(14 - ex) and 2^(ex - 21) are normal;
// mx = 0 gives 0, inf / NaN give 0 as well: the row's terms are then inf / NaN and so is every product (loud)
__device__ __forceinline__ int row_exponent(float mx) {
  const int ex = __builtin_amdgcn_frexp_expf(mx);
  return ex < -100 ? -100 : (ex > 100 ? 100 : ex);
}

// TILES feature tiles of this wave x the 16 rows in LDS, contraction over `chunks` k-chunks of 32:
//   acc[j] += A[tile j][0 .. 32 chunks) . B        (units: 2^kImgScaleExp x the rows' units)
// Two operand sets (one chunk each) in flight: the loads of one are issued, then the MFMAs of the other run — the
// sched_barriers pin that order.  `start` issues the first set: callers run it BEFORE the barrier that publishes B.
template <int TILES>
struct ImgLayer {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned toff[TILES];
  int chunks;
  img_u32x4 wa[TILES][2];

  __device__ __forceinline__ void fill(img_u32x4 (&w)[TILES][2], int c) const {
    const unsigned block = (unsigned)min(c, chunks - 1) * (2u * kImgTermBytes);     // past the end: re-read, unused
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
      w[j][0] = image16(rsrc, toff[j], block);
      w[j][1] = image16(rsrc, toff[j], block + kImgTermBytes);
    }
  }
  __device__ __forceinline__ void start(const char* image, int chunks_, const int (&tile)[TILES], int lane) {
    rsrc = image_buffer(image);
    chunks = chunks_;
#pragma unroll
    for (int j = 0; j < TILES; ++j)
      toff[j] = (unsigned)tile[j] * (unsigned)chunks * (2u * kImgTermBytes) + 16u * (unsigned)lane;
    fill(wa, 0);
  }
  // bhi / blo: this lane's B rows (row m, halfs 8 g ...) of the two terms
  __device__ __forceinline__ void run(f32x4 (&acc)[TILES], const _Float16* bhi, const _Float16* blo) {
    img_u32x4 wb[TILES][2];
    auto compute = [&](const img_u32x4 (&w)[TILES][2], int c) {
      const img_u32x4 bh = *reinterpret_cast<const img_u32x4*>(bhi + 32 * c);
      const img_u32x4 bl = *reinterpret_cast<const img_u32x4*>(blo + 32 * c);
#pragma unroll
      for (int j = 0; j < TILES; ++j) acc[j] = mfma_h16(w[j][1], bh, acc[j]);
#pragma unroll
      for (int j = 0; j < TILES; ++j) acc[j] = mfma_h16(w[j][0], bl, acc[j]);
#pragma unroll
      for (int j = 0; j < TILES; ++j) acc[j] = mfma_h16(w[j][0], bh, acc[j]);
    };
    for (int c = 0; c < chunks; c += 2) {
      fill(wb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(wa, c);
      __builtin_amdgcn_sched_barrier(0);
      fill(wa, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < chunks) compute(wb, c + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// What a hidden layer's epilogue does with the 16 values v[j][e] of this lane (tile j, feature 16 tile + 4 g + e,
// row m): the fp32 row to HBM (weight gradients), and — `operand` — the row as the NEXT product's B operand:
// the row's maximum over all features (lane groups by shuffles, the four waves through LDS), its unit, the
// hi / lo terms into dst (pitch pd halfs), columns [H, pad32(H)) zeroed, the row's descale factor
// 2^(ex - 14 - 7) into desc[slot][row].  Two workgroup barriers; all threads call it.
struct ImgPublish {
  float* rowmax; float* desc;           // LDS: [4 waves][16], [2 slots][16]
  int wave, m, g, tiles, H, r0;
  bool row_ok;
  const int* tile_of;
};
__device__ __forceinline__ void img_publish(const ImgPublish& p, const float (&v)[kMaxTiles][4], float* global,
                                            int ld, bool operand, _Float16* dst_hi, _Float16* dst_lo, int pd,
                                            int slot) {
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    if (p.wave + 4 * j >= p.tiles) break;               // scalar
    const int f = 16 * p.tile_of[j] + 4 * p.g;
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[j][e]));
    if (p.row_ok && global != nullptr)                  // (null: nobody reads this row from HBM)
      *reinterpret_cast<f32x4_dword*>(global + (int64_t)(p.r0 + p.m) * ld + f) =
          f32x4_dword{v[j][0], v[j][1], v[j][2], v[j][3]};
  }
  if (!operand) return;                                 // scalar
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (p.g == 0) p.rowmax[16 * p.wave + p.m] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(p.rowmax[p.m], p.rowmax[16 + p.m]), fmaxf(p.rowmax[32 + p.m], p.rowmax[48 + p.m]));
  const int ex = row_exponent(mx);
  const float unit = img_pow2(kActTop - ex);
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    if (p.wave + 4 * j >= p.tiles) break;
    const int f = 16 * p.tile_of[j] + 4 * p.g;
    unsigned h0, l0, h1, l1;
    img_split(v[j][0] * unit, v[j][1] * unit, h0, l0);
    img_split(v[j][2] * unit, v[j][3] * unit, h1, l1);
    *reinterpret_cast<img_u32x2*>(dst_hi + p.m * pd + f) = img_u32x2{h0, h1};
    *reinterpret_cast<img_u32x2*>(dst_lo + p.m * pd + f) = img_u32x2{l0, l1};
  }
  if (p.wave == 0) {
    if (p.g == 0) p.desc[16 * slot + p.m] = img_pow2(ex - kActTop - kImgScaleExp);
    if (pad32(p.H) != p.H) {                           // (H = 16 mod 32: the upper half of the last k-chunk)
      _Float16* dst = (p.g & 1) ? dst_lo : dst_hi;
      *reinterpret_cast<img_u32x4*>(dst + p.m * pd + p.H + 8 * (p.g >> 1)) = img_u32x4{0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------- forward
__device__ __forceinline__ void mlp_forward_body_img(const MlpFwdArgs& a, const int net, const int bx,
                                                     float* lds_f) {
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int H = a.H, tiles = H / 16, K1 = a.K1;
  const int KP1 = pad32(K1);
  const ImgLds L = img_lds(K1, H);
  const int r0 = bx * kRows;
  const bool row_ok = r0 + m < a.B;
  const bool second = net >= a.split;                 // scalar
  const int post = second ? a.tail2.post : a.post;
  const int64_t poff = net * a.stride_params + (second ? a.second_params : 0);
  const char* images = a.img.block + net * a.img.stride + (second ? a.img.second : 0);
  // developer probe (tonic_debug_forward_stamps): wall-clock stamps (10 ns ticks) of workgroup (0, 0)
  const bool probe = a.stamps != nullptr && bx == 0 && net == 0 && tid == 0;
  auto stamp = [&](int i) {
    if (probe) {
      __builtin_amdgcn_sched_barrier(0);
      a.stamps[i] = wall_clock64();
      a.stamps[8 + i] = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  const float* b1 = a.b1 + poff;
  const float* b2 = a.b2 + poff;
  float* h1g = net >= a.hidden_from ? a.h1 + net * a.stride_hidden : nullptr;      // (scalar)
  float* h2g = net >= a.hidden_from ? a.h2 + net * a.stride_hidden : nullptr;
  _Float16* A_hi = reinterpret_cast<_Float16*>(lds + L.off_a);
  _Float16* A_lo = A_hi + kRows * L.pa;
  _Float16* B_hi = reinterpret_cast<_Float16*>(lds + L.off_b);
  _Float16* B_lo = B_hi + kRows * L.pb;
  float* rowmax = reinterpret_cast<float*>(lds + L.off_misc);
  float* partial = rowmax + 64;
  float* desc = partial + 64;
  if (a.reset_area != nullptr) {
    // the launch AHEAD of the chained ones empties their exchange area (kExchangeEmpty everywhere)
    const int row_tiles = (a.B + kRows - 1) / kRows;
    const int64_t first = ((int64_t)net * (a.reset_blocks > 0 ? row_tiles : (int)gridDim.x) + bx) * blockDim.x + tid;
    const int64_t stride = (int64_t)(a.reset_blocks > 0 ? a.reset_blocks : (int)(gridDim.x * gridDim.y)) * blockDim.x;
    unsigned* area = reinterpret_cast<unsigned*>(a.reset_area);
    for (int64_t i = first; i < a.reset_floats; i += stride) area[i] = kExchangeEmpty;
    if (first == 0 && a.reset_failed != nullptr) *a.reset_failed = 0u;
  }
  int tile_of[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) tile_of[j] = min(wave + 4 * j, tiles - 1);

  // The input rows are requested FIRST (thread = (row, 4-column slot)): a wave's loads return in issue order, and
  // the split below waits for these — behind the weight blocks and biases they cost 1 us more per pass
  // (profiles/r06_forward_stamps_images.txt: inputs -> LDS 2.5 us against 1.4 us of entry for the float32 pass).
  const int prow = tid >> 4, slot = tid & 15;
  const float* xrow = (second ? a.X2 : a.X) + (int64_t)min(r0 + prow, a.B - 1) * a.ldx;
  const int U = (KP1 + 63) / 64;                      // scalar: 64-column strips
  auto element = [&](int c) { const float v = xrow[min(c, K1 - 1)]; return c < K1 ? v : 0.f; };
  float xv[4][4];                                     // (K1 <= 256: the row's share stays in registers)
  if (U <= 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u >= U) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[u][e] = element(64 * u + 4 * slot + e);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // layer 1's first weight blocks, then everything else that does not depend on activations
  ImgLayer<kMaxTiles> l1;
  l1.start(images + a.img.f1.off, a.img.f1.chunks, tile_of, lane);
  f32x4 bias1[kMaxTiles], bias2[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    const f32x4_dword q1 = *reinterpret_cast<const f32x4_dword*>(b1 + 16 * tile_of[j] + 4 * g);
    const f32x4_dword q2 = *reinterpret_cast<const f32x4_dword*>(b2 + 16 * tile_of[j] + 4 * g);
    bias1[j] = f32x4{q1[0], q1[1], q1[2], q1[3]};
    bias2[j] = f32x4{q2[0], q2[1], q2[2], q2[3]};
  }
  const int tiles_per_head = (a.NH + 15) / 16;
  const bool value_head = a.heads == 1 && a.NH == 1 && a.act[0] == ACT_NONE;       // scalar
  const bool head_wave = !value_head && wave < a.heads * tiles_per_head;
  const int head = head_wave ? wave / tiles_per_head : 0;
  const int head_tile = head_wave ? wave - head * tiles_per_head : 0;
  const float* Wh = (head == 0 ? a.Wh[0] : a.Wh[1]) + poff;
  const float* bh = (head == 0 ? a.bh[0] : a.bh[1]) + poff;
  f32x4 wvalue[kMaxTiles];                            // value head: w3 of this lane's features (fp32, VALU)
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j)
    wvalue[j] = value_head ? load_w4(Wh, 16 * tile_of[j] + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
  float hbias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) hbias[e] = bh[min(16 * head_tile + 4 * g + e, a.NH - 1)];

  // ---- the 16 input rows -> region A as hi / lo terms in each row's own unit
  {
    auto store4 = [&](int c0, float x0, float x1, float x2, float x3, float unit) {
      unsigned h0, l0, h1, l1;
      img_split(x0 * unit, x1 * unit, h0, l0);
      img_split(x2 * unit, x3 * unit, h1, l1);
      *reinterpret_cast<img_u32x2*>(A_hi + prow * L.pa + c0) = img_u32x2{h0, h1};
      *reinterpret_cast<img_u32x2*>(A_lo + prow * L.pa + c0) = img_u32x2{l0, l1};
    };
    int ex;
    if (U <= 4) {
      float amax = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u >= U) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(xv[u][e]));
      }
      ex = row_exponent(img_row_max16(amax));
      const float unit = img_pow2(kActTop - ex);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u >= U) break;
        const int c0 = 64 * u + 4 * slot;
        if (c0 < KP1) store4(c0, xv[u][0], xv[u][1], xv[u][2], xv[u][3], unit);
        if (a.rows_out != nullptr && r0 + prow < a.B) {       // (scalar pointer: a collector step's device copy)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < K1) a.rows_out[(int64_t)(r0 + prow) * a.rows_ld + c0 + e] = xv[u][e];
        }
      }
    } else {                                          // wide inputs: two passes over the row (L2 hits)
      float amax = 0.f;
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(element(64 * u + 4 * slot + e)));
      }
      ex = row_exponent(img_row_max16(amax));
      const float unit = img_pow2(kActTop - ex);
      for (int u = 0; u < U; ++u) {
        const int c0 = 64 * u + 4 * slot;
        const float x0 = element(c0), x1 = element(c0 + 1), x2 = element(c0 + 2), x3 = element(c0 + 3);
        if (c0 < KP1) store4(c0, x0, x1, x2, x3, unit);
        if (a.rows_out != nullptr && r0 + prow < a.B) {
          const float xs[4] = {x0, x1, x2, x3};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < K1) a.rows_out[(int64_t)(r0 + prow) * a.rows_ld + c0 + e] = xs[e];
        }
      }
    }
    if (slot == 0) desc[prow] = img_pow2(ex - kActTop - kImgScaleExp);
  }
  __syncthreads();
  stamp(1);                                           // input rows fetched, split, published

  const ImgPublish pub{rowmax, desc, wave, m, g, tiles, H, r0, row_ok, tile_of};
  f32x4 acc[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  l1.run(acc, A_hi + m * L.pa + 8 * g, A_lo + m * L.pa + 8 * g);
  stamp(2);                                           // layer 1's products
  ImgLayer<kMaxTiles> l2;
  l2.start(images + a.img.f2.off, a.img.f2.chunks, tile_of, lane);     // W2's first blocks fly over the epilogue
  {
    const float dsc = desc[m];
    float h[kMaxTiles][4];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) h[j][e] = fmaxf(fmaf(acc[j][e], dsc, bias1[j][e]), 0.f);
    img_publish(pub, h, h1g, a.ldh, true, B_hi, B_lo, L.pb, 1);
  }
  stamp(3);                                           // layer 1's epilogue: h1 to HBM, unit, split, two barriers

#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  l2.run(acc, B_hi + m * L.pb + 8 * g, B_lo + m * L.pb + 8 * g);
  stamp(4);                                           // layer 2's products
  // a head tile is one short chain on one wave: all of its weight blocks are requested here, over the second
  // layer's epilogue and its barriers
  img_u32x4 wh[kImgMaxChunks][2];
  if (head_wave) {
    const __amdgpu_buffer_rsrc_t hr = image_buffer(images + (head == 0 ? a.img.fh[0].off : a.img.fh[1].off));
    const int hc = a.img.fh[0].chunks;
    const unsigned toff = (unsigned)head_tile * (unsigned)hc * (2u * kImgTermBytes) + 16u * (unsigned)lane;
#pragma unroll
    for (int c = 0; c < kImgMaxChunks; ++c) {
      const unsigned block = (unsigned)min(c, hc - 1) * (2u * kImgTermBytes);
      wh[c][0] = image16(hr, toff, block);
      wh[c][1] = image16(hr, toff, block + kImgTermBytes);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float h2[kMaxTiles][4];
  {
    const float dsc = desc[16 + m];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) h2[j][e] = fmaxf(fmaf(acc[j][e], dsc, bias2[j][e]), 0.f);
  }
  if (value_head) {
    // q[row] = b3 + sum_f h2[row][f] * w3[f] on the fp32 activations: this lane's 16 features, the four lane
    // groups of the row, the four waves through LDS
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      if (wave + 4 * j >= tiles) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) part += h2[j][e] * wvalue[j][e];
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (g == 0) partial[16 * wave + m] = part;
    img_publish(pub, h2, h2g, a.ldh, false, nullptr, nullptr, 0, 0);
    __syncthreads();
    if (wave == 0 && g == 0 && row_ok) {
      float* out_base = a.out[0];
      const float q = ((partial[m] + partial[16 + m]) + (partial[32 + m] + partial[48 + m])) + hbias[0];
      out_base[net * a.stride_out + (int64_t)(r0 + m) * a.ldo] = q;
      if (a.xq != nullptr) {       // read by other workgroups of the same launch (ValueLines)
        const int slot = net < a.split ? 32 * net : 32 * (2 + net - a.split);     // one line per writer
        exchange_write(a.xq + (int64_t)bx * kExchangeTileFloats + slot + m, q);
      }
    }
    stamp(5);                                         // value head out
    return;
  }
  img_publish(pub, h2, h2g, a.ldh, true, A_hi, A_lo, L.pa, 0);
  stamp(5);                                           // layer 2's epilogue

  // heads: one [16 outputs][16 rows] tile per head wave
  if (head_wave) {
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    const _Float16* bhi = A_hi + m * L.pa + 8 * g;
    const _Float16* blo = A_lo + m * L.pa + 8 * g;
    const int hc = a.img.fh[0].chunks;
#pragma unroll
    for (int c = 0; c < kImgMaxChunks; ++c) {
      if (c < hc) {                                   // scalar
        const img_u32x4 vh = *reinterpret_cast<const img_u32x4*>(bhi + 32 * c);
        const img_u32x4 vl = *reinterpret_cast<const img_u32x4*>(blo + 32 * c);
        out = mfma_h16(wh[c][1], vh, out);
        out = mfma_h16(wh[c][0], vl, out);
        out = mfma_h16(wh[c][0], vh, out);
      }
    }
    const float dsc = desc[m];
    float* out_base = head == 0 ? a.out[0] : a.out[1];
    const int act = head == 0 ? a.act[0] : a.act[1];
    float* dst = out_base + net * a.stride_out + (int64_t)(r0 + m) * a.ldo;
    float* headbuf = reinterpret_cast<float*>(lds + L.off_f32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = 16 * head_tile + 4 * g + e;
      if (o < a.NH) {
        float v = fmaf(out[e], dsc, hbias[e]);
        if (act == ACT_TANH) v = tanhf(v);
        if (row_ok) dst[o] = v;
        if (post != POST_NONE) headbuf[(head * kRows + m) * kPostPitch + o] = v;
      }
    }
  }
  stamp(6);                                           // heads
  if (post == POST_NONE) return;                    // scalar
  policy_tail(a, second, r0, lds_f);                  // (a.tail_offset = L.off_f32 / 4: set by the launcher)
  stamp(7);
}

// ------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ void mlp_backward_body_img(const MlpBwdArgs& a, const int net, const int bx,
                                                      float* lds_f, const int K1_lds) {
  char* lds = reinterpret_cast<char*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int H = a.H, tiles = H / 16;
  const ImgLds L = img_lds(K1_lds, H);                // (the forward's layout when both run in one launch)
  const int r0 = bx * kRows;
  unsigned* co = a.exchange_failed;                   // scalar; null: plain loads / stores
  const int row = min(r0 + m, a.B - 1);
  const bool row_ok = r0 + m < a.B;
  const char* images = a.img.block + net * a.img.stride;
  const float* h1g = a.h1 + net * a.stride_hidden;
  const float* h2g = a.h2 + net * a.stride_hidden;
  float* dz2g = a.skip_dz ? nullptr : a.dz2 + net * a.stride_hidden;                 // (scalar)
  float* dz1g = a.skip_dz ? nullptr : a.dz1 + net * a.stride_hidden;
  _Float16* A_hi = reinterpret_cast<_Float16*>(lds + L.off_a);
  _Float16* A_lo = A_hi + kRows * L.pa;
  _Float16* B_hi = reinterpret_cast<_Float16*>(lds + L.off_b);
  _Float16* B_lo = B_hi + kRows * L.pb;
  float* rowmax = reinterpret_cast<float*>(lds + L.off_misc);
  float* desc = rowmax + 128;
  float* dhl = reinterpret_cast<float*>(lds + L.off_f32);      // [2 heads][16 rows][kHeadPitch] (formed only)
  int tile_of[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) tile_of[j] = min(wave + 4 * j, tiles - 1);

  // The folded head backward, part 1: its operands through clamped addresses, no branch around a load
  const bool formed = a.heads > 0 && a.hb_dxa0 != nullptr;      // scalar
  constexpr int kHeadSlots = kPostPitch / 16;
  float hb_da[kHeadSlots], hb_second[kHeadSlots], hb_t[kHeadSlots], hb_sg[kHeadSlots], hb_ep[kHeadSlots],
      hb_pre[kHeadSlots];
  const int hb_row = tid >> 4, hb_slot = tid & 15;
  if (formed) {
    const int A = a.NH;
    const int64_t src = min((int64_t)r0 + hb_row, (int64_t)a.B - 1);
    const float* dxa1 = a.hb_dxa1 != nullptr ? a.hb_dxa1 : a.hb_dxa0;
    const float* sgp = a.hb_sac ? a.hb_sigma : a.hb_act;
    const float* epp = a.hb_sac ? a.hb_eps : a.hb_act;
    const float* prep = a.hb_sac ? a.hb_spre : a.hb_act;
    const int64_t pre_ld = a.hb_sac ? a.ldh : A;
#pragma unroll
    for (int u = 0; u < kHeadSlots; ++u) {
      const int aa = min(hb_slot + 16 * u, A - 1);
      if (co == nullptr) {                            // (scalar; chained: further down)
        hb_da[u] = a.hb_dxa0[src * a.hb_ldxa + aa];
        hb_second[u] = dxa1[src * a.hb_ldxa + aa];
      }
      hb_t[u] = a.hb_act[src * A + aa];
      hb_sg[u] = sgp[src * A + aa];
      hb_ep[u] = epp[src * A + aa];
      hb_pre[u] = prep[src * pre_ld + aa];
    }
  }
  // ReLU masks of both layers (forward activations of this lane's row / features) and the first blocks of W2^T
  f32x4 mask2[kMaxTiles], mask1[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    const int f = 16 * tile_of[j] + 4 * g;
    mask2[j] = load_k4(h2g + (int64_t)row * a.ldhid, f);
    mask1[j] = load_k4(h1g + (int64_t)row * a.ldhid, f);
  }
  ImgLayer<kMaxTiles> l1;                             // dz1 = W2^T dz2: requested before dz2 exists
  l1.start(images + a.img.t2.off, a.img.t2.chunks, tile_of, lane);
  // actor: the heads' transposed images are short (pad32(A) / 32 = one or two k-chunks per head): all of their
  // blocks are requested here, ahead of the head backward's arithmetic and barrier
  const int hchunks = a.heads > 0 ? a.img.th[0].chunks : 0;      // scalar
  img_u32x4 whd[2][2][kMaxTiles][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h >= a.heads) break;                          // scalar
    const __amdgpu_buffer_rsrc_t hr = image_buffer(images + (h == 0 ? a.img.th[0].off : a.img.th[1].off));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c >= hchunks) break;                        // scalar
#pragma unroll
      for (int j = 0; j < kMaxTiles; ++j) {
        const unsigned toff = (unsigned)tile_of[j] * (unsigned)hchunks * (2u * kImgTermBytes) + 16u * (unsigned)lane;
        whd[h][c][j][0] = image16(hr, toff, (unsigned)c * (2u * kImgTermBytes));
        whd[h][c][j][1] = image16(hr, toff, (unsigned)c * (2u * kImgTermBytes) + kImgTermBytes);
      }
    }
  }

  if (co != nullptr && formed) {                      // scalar
    const int A = a.NH;
    const int64_t src = min((int64_t)r0 + hb_row, (int64_t)a.B - 1);
    const float* dxa1 = a.hb_dxa1 != nullptr ? a.hb_dxa1 : a.hb_dxa0;
    const float* where[2 * kHeadSlots];
    float got[2 * kHeadSlots];
#pragma unroll
    for (int u = 0; u < kHeadSlots; ++u) {
      const int aa = min(hb_slot + 16 * u, A - 1);
      where[2 * u] = a.hb_dxa0 + src * a.hb_ldxa + aa;
      where[2 * u + 1] = dxa1 + src * a.hb_ldxa + aa;
    }
    shared_values(where, got, co);                   // (all in flight together: one round trip, not eight)
#pragma unroll
    for (int u = 0; u < kHeadSlots; ++u) { hb_da[u] = got[2 * u]; hb_second[u] = got[2 * u + 1]; }
  }
  if (formed && a.hb_dxa1 != nullptr) {
#pragma unroll
    for (int u = 0; u < kHeadSlots; ++u) hb_da[u] = hb_da[u] + hb_second[u];
  }
  const ImgPublish pub{rowmax, desc, wave, m, g, tiles, H, r0, row_ok, tile_of};
  float d2[kMaxTiles][4];
  if (a.heads == 0) {                                 // critic: dq[row] * w3[feature], fp32
    float dq;
    if (a.loss == LOSS_GIVEN) {
      dq = a.dq[net * a.stride_dq + row];
    } else {                                          // the step's loss, folded into this launch
      if (a.loss == LOSS_TD) {
        const float y = td_target(a.l_rewards, a.l_discounts, a.l_tq, a.l_logp, a.l_alpha, row, a.l_tq_at,
                                  a.l_nets, co);
        dq = 2.f * (load_shared(a.l_q + a.l_q_at.index(net, row), co) - y);
      } else {
        dq = actor_dq(a.l_q, row, a.l_q_at, a.l_nets == 2, net, co);
      }
      if (row_ok && wave == 0 && g == 0)              // for the weight-gradient GEMM (dw3, db3)
        const_cast<float*>(a.dq)[net * a.stride_dq + row] = dq;
    }
    const float* w3 = a.w3 + net * a.stride_params;
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      const f32x4 w = load_w4(w3, 16 * tile_of[j] + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) d2[j][e] = mask2[j][e] > 0.f ? dq * w[e] : 0.f;
    }
  } else {                                            // actor: sum over the heads of Wh^T dhead
    const int A = a.NH;
    if (formed) {
      // part 2: actor_head_backward_kernel for the 16 rows of this workgroup, thread = (row, action slot): the
      // same expressions; -> LDS (fp32; the head products' B operand is split from there) and -> dhead[.] in HBM
      const int64_t grow = r0 + hb_row;
      const bool ok = grow < a.B;
#pragma unroll
      for (int u = 0; u < kHeadSlots; ++u) {
        const int aa = hb_slot + 16 * u;
        float dloc = 0.f, dspre = 0.f;
        if (aa < A) {
          const float da = hb_da[u], t = hb_t[u];
          const float one_m = 1.f - t * t;
          if (!a.hb_sac) {
            dloc = da * one_m;
          } else {
            const float du = da * one_m + a.hb_alpha * (2.f * t * one_m / (one_m + kSacLogEps));
            const float dsigma = du * hb_ep[u] - a.hb_alpha / hb_sg[u];
            const float pre = hb_pre[u];
            const float raw = softplus_f(pre);
            const bool inside = raw >= 1e-4f && raw <= 1.0f;
            dloc = du;
            dspre = inside ? dsigma / (1.f + expf(-pre)) : 0.f;
          }
          if (ok) {
            const_cast<float*>(a.dhead[0])[grow * a.ldh + aa] = dloc;
            if (a.hb_sac) const_cast<float*>(a.dhead[1])[grow * a.ldh + aa] = dspre;
          }
        }
        dhl[hb_row * kHeadPitch + aa] = dloc;
        dhl[(kRows + hb_row) * kHeadPitch + aa] = dspre;
      }
      __syncthreads();
    }
    // this lane's share of the B operand: row m, actions 32 c + 8 g + e of head h — from LDS (formed) or HBM
    // (given), zero beyond the head's width; ONE unit per row over both heads (the products of the heads add up)
    float dv[2][2][8];                                // (hchunks = pad32(A) / 32: one or two)
    float mx = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= a.heads) break;                        // scalar
      const float* given = (h == 0 ? a.dhead[0] : a.dhead[1]) + (int64_t)row * a.ldh;
      const float* made = dhl + (h * kRows + m) * kHeadPitch;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c >= hchunks) break;                      // scalar
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * c + 8 * g + e;
          const float v = formed ? made[min(k, kPostPitch - 1)] : given[min(k, A - 1)];
          dv[h][c][e] = k < A ? v : 0.f;
          mx = fmaxf(mx, fabsf(dv[h][c][e]));
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const int ex = row_exponent(mx);
    const float unit = img_pow2(kActTop - ex), dsc = img_pow2(ex - kActTop - kImgScaleExp);
    f32x4 acc[kMaxTiles];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= a.heads) break;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c >= hchunks) break;
        img_u32x4 bh, bl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned hi, lo;
          img_split(dv[h][c][2 * p] * unit, dv[h][c][2 * p + 1] * unit, hi, lo);
          bh[p] = hi; bl[p] = lo;
        }
#pragma unroll
        for (int j = 0; j < kMaxTiles; ++j) acc[j] = mfma_h16(whd[h][c][j][1], bh, acc[j]);
#pragma unroll
        for (int j = 0; j < kMaxTiles; ++j) acc[j] = mfma_h16(whd[h][c][j][0], bl, acc[j]);
#pragma unroll
        for (int j = 0; j < kMaxTiles; ++j) acc[j] = mfma_h16(whd[h][c][j][0], bh, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) d2[j][e] = mask2[j][e] > 0.f ? acc[j][e] * dsc : 0.f;
  }
  img_publish(pub, d2, dz2g, a.ldhid, true, A_hi, A_lo, L.pa, 0);

  f32x4 acc[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  l1.run(acc, A_hi + m * L.pa + 8 * g, A_lo + m * L.pa + 8 * g);
  // the action columns of the input gradient: one 16-column tile per wave, its blocks requested over the epilogue
  const int xa_tiles = (a.xa_count + 15) / 16;
  const bool xa_wave = wave < xa_tiles;               // scalar
  img_u32x4 wx[kImgMaxChunks][2];
  if (xa_wave) {
    const __amdgpu_buffer_rsrc_t xr = image_buffer(images + a.img.t1a.off);
    const int xc = a.img.t1a.chunks;
    const unsigned toff = (unsigned)wave * (unsigned)xc * (2u * kImgTermBytes) + 16u * (unsigned)lane;
#pragma unroll
    for (int c = 0; c < kImgMaxChunks; ++c) {
      const unsigned block = (unsigned)min(c, xc - 1) * (2u * kImgTermBytes);
      wx[c][0] = image16(xr, toff, block);
      wx[c][1] = image16(xr, toff, block + kImgTermBytes);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float dsc = desc[m];
    float d1[kMaxTiles][4];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) d1[j][e] = mask1[j][e] > 0.f ? acc[j][e] * dsc : 0.f;
    img_publish(pub, d1, dz1g, a.ldhid, a.xa_count > 0, B_hi, B_lo, L.pb, 1);
  }
  if (xa_wave) {                                      // [16 action columns][16 rows] of dz1 . W1
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    const _Float16* bhi = B_hi + m * L.pb + 8 * g;
    const _Float16* blo = B_lo + m * L.pb + 8 * g;
    const int xc = a.img.t1a.chunks;
#pragma unroll
    for (int c = 0; c < kImgMaxChunks; ++c) {
      if (c < xc) {
        const img_u32x4 vh = *reinterpret_cast<const img_u32x4*>(bhi + 32 * c);
        const img_u32x4 vl = *reinterpret_cast<const img_u32x4*>(blo + 32 * c);
        out = mfma_h16(wx[c][1], vh, out);
        out = mfma_h16(wx[c][0], vl, out);
        out = mfma_h16(wx[c][0], vh, out);
      }
    }
    const float dsc = desc[16 + m];
    if (row_ok) {
      float* dst = a.dxa + net * a.stride_dxa + (int64_t)(r0 + m) * a.ldxa;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = 16 * wave + 4 * g + e;
        if (o < a.xa_count) {
          if (co != nullptr) exchange_write(dst + o, out[e] * dsc);
          else dst[o] = out[e] * dsc;
        }
      }
    }
  }
  // the logged sums (uniform); a chain launch leaves them to its last workgroup
  if (a.loss != LOSS_GIVEN && co == nullptr && bx == 0 && net == 0) mlp_loss_stats(a);
}
