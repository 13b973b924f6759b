// Fused forward of a two-hidden-layer ReLU MLP with up to two narrow heads (gfx950) — the
// networks of the off-policy path: Actor / Critic of tonic/torch/models/{actors,critics}.py with
// the MLP torso of models/utils.py:12-23 (x -> relu(W1 x + b1) -> relu(W2 h1 + b2) -> heads).
//
// At batch 1024 and width 256 one layer is 134 MFLOP: as separate GEMM launches every layer pays
// the ~5 us a dependent launch costs (first loads of a fresh kernel + store + launch), three to
// four times per network pass.  Here one workgroup of four waves carries 16 batch rows through all
// layers: wave w owns the 16-feature output tiles w, w+4, w+8, w+12 of both hidden layers
// (v_mfma_f32_16x16x4_f32, product formed as D[feature][row] so that the result is already in
// the lane = row layout the next layer wants as B operand), hidden activations are exchanged
// through LDS, weights stream from L2 as one aligned 16-byte load per lane and k-chunk (rows are
// weight_ld floats apart, see mlpfwd.h), and the hidden
// activations are also written to HBM because the backward pass needs them.
#include "mlpfwd.h"

#include <atomic>
#include <vector>

namespace tonic {

namespace {

typedef float f32x4_dword __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four consecutive k of one row as one dwordx4 load (dword alignment is enough on gfx950).
__device__ __forceinline__ f32x4 load_k4(const float* row, int k) {
  const f32x4_dword q = *reinterpret_cast<const f32x4_dword*>(row + k);
  return f32x4{q[0], q[1], q[2], q[3]};
}

// Weight rows are 16-byte aligned (weight_ld).
__device__ __forceinline__ f32x4 load_w4(const float* row, int k) {
  return *reinterpret_cast<const f32x4*>(row + k);
}

// The ragged last chunk (K % 16 != 0): clamped scalar loads, k >= K zeroed by the caller's mask.
__device__ __forceinline__ f32x4 load_k4_tail(const float* row, int k, int K) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = row[min(k + e, K - 1)];
  return v;
}

constexpr int kRows = 16;          // batch rows per workgroup
constexpr int kPostPitch = 64;     // head outputs per row in the policy tail (NH <= 64)
constexpr int kHeadPitch = kPostPitch + 4;   // LDS rows of the folded head backward (16-byte aligned)
constexpr int kMaxTiles = 4;       // 16-feature tiles per wave: H <= 256

// One layer for the TILES 16-feature tiles a wave owns (rows wrow[j] of W), contraction over K.
//   acc[j] += W[tile j rows][0 .. K) . B operand
// bfull(k) / btail(k) give the B operand of this lane: four consecutive k of its input row.
// Only one wave lives on a SIMD here, so latency is hidden by software: two operand sets of
// kHalf k-chunks each; the loads of one set are issued, THEN the MFMAs of the other set run (the
// sched_barriers pin that order — left alone the scheduler sinks every load next to its use and
// the loop becomes "wait for everything, compute, load everything").  The loop has no branch but
// its back edge: chunks past the end re-read the last chunk and are multiplied by zero.  The
// weight half of the first set and the ragged tail are requested by `start`, which the caller
// runs BEFORE the barrier that publishes the previous layer's activations.
constexpr int kHalf = 2;

// KS = false: the weight rows are k-contiguous (forward: W[out][k]); KS = true: k-strided
// (backward through W^T: element (out, k) lives at W[k][out], i.e. base + k * ld + out).
//
// Addressing: BUFFER loads — the tensor as a buffer resource (four SGPRs), off[j] this lane's 32-bit
// byte offset inside it (tile j's row / column + its k group) and the chunk's byte offset as the
// scalar offset operand: a load costs no vector address arithmetic at all.  (Per-lane 64-bit row
// pointers cost a v_lshl_add_u64 per load — also in the `scalar base + zero-extended offset` form,
// once the zero-extension has been hoisted out of the loop — and VALU instructions do not overlap
// fp32 MFMAs on gfx950: ~50 of them per 64 MFMAs of the layer loops, profiles/r03_offpolicy_phases.md.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_buffer(const float* tensor) {
  // raw buffer (stride 0), no bound in practice; 0x00020000: 32-bit float data format (gfx9)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tensor), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 buffer4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned chunk_bytes) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, chunk_bytes, 0));
}
__device__ __forceinline__ float buffer1(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned chunk_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, chunk_bytes, 0));
}

template <int TILES, bool KS = false>
struct Layer {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned off[TILES];           // BYTES: !KS 4 (row_j * ld + 4 kg);  KS 4 (4 kg * ld + column_j)
  int K, nfull, ld, kg;

  // chunk c (k = 16 c .. 16 c + 15) of tile j: this lane's four k
  __device__ __forceinline__ f32x4 weights(int j, int c) const {
    if (!KS) return buffer4(rsrc, off[j], 64u * (unsigned)c);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = buffer1(rsrc, off[j], 4u * (unsigned)((16 * c + e) * ld));
    return v;
  }
  __device__ __forceinline__ f32x4 weights_tail(int j) const {
    // (k-contiguous rows are padded to a multiple of 4 floats: the chunk is clamped into the
    //  row, whatever lies at k >= K meets a zeroed B operand)
    const int k = 16 * nfull + 4 * kg;
    if (!KS) return buffer4(rsrc, off[j] + 4u * (unsigned)(min(k, (K + 3) / 4 * 4 - 4) - 4 * kg), 0u);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = buffer1(rsrc, off[j] + 4u * (unsigned)((min(k + e, K - 1) - 4 * kg) * ld), 0u);
    return v;
  }

  f32x4 a0[kHalf][TILES];        // weight operands of the first set
  f32x4 at[TILES];               // weight operands of the ragged last chunk (K % 16 != 0)

  // lane_first[j]: !KS the first element of tile j's row of this lane (row * ld); KS its column
  __device__ __forceinline__ void start(const float* tensor, const unsigned (&lane_first)[TILES],
                                        int K_, int kg_, int ld_) {
    rsrc = weight_buffer(tensor); K = K_; nfull = K / 16; ld = ld_; kg = kg_;
#pragma unroll
    for (int j = 0; j < TILES; ++j) off[j] = 4u * (lane_first[j] + (KS ? 4 * kg * ld : 4 * kg));
#pragma unroll
    for (int q = 0; q < kHalf; ++q) {
      const int c = min(q, max(nfull - 1, 0));
#pragma unroll
      for (int j = 0; j < TILES; ++j) a0[q][j] = nfull > 0 ? weights(j, c) : f32x4{0, 0, 0, 0};
    }
    if (K % 16 != 0) {
#pragma unroll
      for (int j = 0; j < TILES; ++j) at[j] = weights_tail(j);
    }
  }

  // Whole rounds of four chunks — two operand sets, the loads of one issued before the MFMAs of the
  // other — then the one to three chunks that are left, straight-line under scalar conditions.
  // (Round 2 ran ceil(nfull / 4) rounds and multiplied the B operands of the re-read chunks by
  // zero: four VALU instructions per chunk in the MFMA stream, which fp32 MFMAs do not hide.)
  template <typename BFull>
  __device__ __forceinline__ void loop(f32x4 (&acc)[TILES], BFull bfull) {
    f32x4 aA[kHalf][TILES], aB[kHalf][TILES], bA[kHalf], bB[kHalf];
    auto fill_a = [&](f32x4 (&a)[kHalf][TILES], int first) {
#pragma unroll
      for (int q = 0; q < kHalf; ++q) {
        const int c = min(first + q, nfull - 1);
#pragma unroll
        for (int j = 0; j < TILES; ++j) a[q][j] = weights(j, c);
      }
    };
    auto fill_b = [&](f32x4 (&b)[kHalf], int first) {
#pragma unroll
      for (int q = 0; q < kHalf; ++q) b[q] = bfull(16 * min(first + q, nfull - 1) + 4 * kg);
    };
    auto chunk = [&](const f32x4 (&a)[TILES], const f32x4& b) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {                 // e outer: the tiles are independent chains
#pragma unroll
        for (int j = 0; j < TILES; ++j) acc[j] = mfma16(a[j][e], b[e], acc[j]);
      }
    };
#pragma unroll
    for (int q = 0; q < kHalf; ++q) {
#pragma unroll
      for (int j = 0; j < TILES; ++j) aA[q][j] = a0[q][j];
    }
    fill_b(bA, 0);
    const int rounds = nfull / (2 * kHalf), rest = nfull % (2 * kHalf);
    for (int r = 0; r < rounds; ++r) {
      const int c = 2 * kHalf * r;
      fill_a(aB, c + kHalf); fill_b(bB, c + kHalf);
      __builtin_amdgcn_sched_barrier(0);
      chunk(aA[0], bA[0]); chunk(aA[1], bA[1]);
      __builtin_amdgcn_sched_barrier(0);
      fill_a(aA, c + 2 * kHalf); fill_b(bA, c + 2 * kHalf);      // (the last round: the rest's first two)
      __builtin_amdgcn_sched_barrier(0);
      chunk(aB[0], bB[0]); chunk(aB[1], bB[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (rest > 0) {                                   // scalar: chunks 4 rounds .. nfull - 1
      if (rest > 2) {
        const int c = nfull - 1;
#pragma unroll
        for (int j = 0; j < TILES; ++j) aB[0][j] = weights(j, c);
        bB[0] = bfull(16 * c + 4 * kg);
      }
      __builtin_amdgcn_sched_barrier(0);
      chunk(aA[0], bA[0]);
      if (rest > 1) chunk(aA[1], bA[1]);
      if (rest > 2) chunk(aB[0], bB[0]);
    }
  }

  template <typename BFull, typename BTail>
  __device__ __forceinline__ void run(f32x4 (&acc)[TILES], BFull bfull, BTail btail) {
    if (nfull > 0) loop(acc, bfull);
    if (K % 16 != 0) {                                // ragged last chunk
      const int k = 16 * nfull + 4 * kg;
      f32x4 b = btail(k);
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = (k + e < K) ? b[e] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int j = 0; j < TILES; ++j) acc[j] = mfma16(at[j][e], b[e], acc[j]);
      }
    }
  }
};

// What follows a policy's heads (sampling / target noise / dense copy, the critics' input rows), for the 16 rows
// of this workgroup: thread = (row, action slot).  The head outputs are in LDS at a.tail_offset
// ([2 heads][16 rows][kPostPitch], written by the head waves).  Shared by the float32 pass and the image pass.
__device__ __forceinline__ void policy_tail(const MlpFwdArgs& a, const bool second, const int r0, float* lds) {
  const int tid = threadIdx.x;
  const int post = second ? a.tail2.post : a.post;
  const float* post_eps = second ? a.tail2.eps : a.post_eps;
  float* post_actions = second ? a.tail2.actions : a.post_actions;
  float* post_sigma = second ? a.tail2.sigma : a.post_sigma;
  float* post_logp = second ? a.tail2.logp : a.post_logp;
  const float* enc_obs = second ? a.tail2.enc_obs : a.enc_obs;
  float* enc_out = second ? a.tail2.enc_out : a.enc_out;
  const float* enc_obs2 = second ? nullptr : a.enc_obs2;
  const float* enc_act2 = second ? nullptr : a.enc_act2;
  float* enc_out2 = second ? nullptr : a.enc_out2;
  // ---- what follows the heads, for the 16 rows of this workgroup: thread = (row, action slot)
  const int prow = tid >> 4, slot = tid & 15, A = a.NH;
  const int64_t grow = r0 + prow;
  const bool ok = grow < a.B;
  // The encoder's operands (raw observations, statistics: lines nobody has touched in this launch)
  // are requested FIRST — they fly over the barrier and the sampling arithmetic below.
  const int O = a.enc_O;
  const int64_t src = min(grow, (int64_t)a.B - 1);
  const bool pair = enc_out2 != nullptr;
  float x[8], y[8], mean[8], sdev[8];
  if (enc_out != nullptr) {                           // scalar
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = min(16 * u + slot, O - 1);
      x[u] = enc_obs[src * O + c];
      y[u] = pair ? enc_obs2[src * O + c] : 0.f;
      mean[u] = a.enc_mean[c];
      sdev[u] = a.enc_std[c];
    }
  }
  // ... and so are this thread's noise draws and stored actions (round 6: loaded inside the action loop, under
  // lane-predicated branches, each waited for where it was used: 1.8 us of the tail's 3.6 for two actions per
  // thread, profiles/r06_forward_stamps_images.txt); clamped addresses, selected below
  float eps4[kPostPitch / 16], act4[kPostPitch / 16];
#pragma unroll
  for (int u = 0; u < kPostPitch / 16; ++u) {
    const int64_t at = src * A + min(slot + 16 * u, A - 1);
    eps4[u] = post_eps != nullptr ? post_eps[at] : 0.f;                         // (scalar conditions)
    act4[u] = (enc_out != nullptr && enc_out2 != nullptr) ? enc_act2[at] : 0.f;
  }
  __syncthreads();
  const float* headbuf = lds + a.tail_offset;         // [2 heads][16 rows][kPostPitch]
  const int padded = (A + 15) / 16 * 16;
  // this thread's log-probability terms: actions slot, slot + 16, slot + 32, slot + 48
  float term4[kPostPitch / 16] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < kPostPitch / 16; ++u) {
    const int aa = slot + 16 * u;
    if (aa >= padded) break;                          // scalar
    float action = 0.f;
    if (aa < A) {
      const float first = headbuf[prow * kPostPitch + aa];
      if (post == POST_SQUASHED_SAMPLE) {
        const bool has_eps = post_eps != nullptr;
        const float eps = (has_eps && ok) ? eps4[u] : 0.f;
        const SquashedSample sm =
            squashed_sample(first, headbuf[(kRows + prow) * kPostPitch + aa], eps, has_eps);
        term4[u] = sm.logp_term;
        if (ok) {
          post_actions[grow * A + aa] = sm.action;
          if (post_sigma != nullptr) post_sigma[grow * A + aa] = sm.sigma;
        }
        action = sm.action;
      } else if (ok) {
        action = post == POST_TARGET_NOISE
                     ? noisy_target_action(first, eps4[u], a.noise_scale, a.noise_clip)
                     : first;
        post_actions[grow * A + aa] = action;
      }
      if (ok && enc_out != nullptr) {               // the critics' input: action columns
        enc_out[grow * a.enc_ld + a.enc_O + aa] = action;
        if (enc_out2 != nullptr)
          enc_out2[grow * a.enc_ld + a.enc_O + aa] = act4[u];
      }
    }
  }
  if (enc_out != nullptr) {
    // ... and the normalised observation columns, eight 16-column strips at a time (the first
    // eight were requested above): all loads first, through clamped addresses (a load under a
    // lane-predicated branch waits for itself)
    for (int c0 = 0; c0 < O; c0 += 8 * 16) {
      if (c0 > 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = min(c0 + 16 * u + slot, O - 1);
          x[u] = enc_obs[src * O + c];
          y[u] = pair ? enc_obs2[src * O + c] : 0.f;
          mean[u] = a.enc_mean[c];
          sdev[u] = a.enc_std[c];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 16 * u + slot;
        if (ok && c < O) {
          enc_out[grow * a.enc_ld + c] =
              __builtin_amdgcn_fmed3f((x[u] - mean[u]) / sdev[u], -a.enc_clip, a.enc_clip);
          if (pair)
            enc_out2[grow * a.enc_ld + c] =
                __builtin_amdgcn_fmed3f((y[u] - mean[u]) / sdev[u], -a.enc_clip, a.enc_clip);
        }
      }
    }
  }
  if (post != POST_SQUASHED_SAMPLE || post_logp == nullptr) return;
  // The fold of sac_sample_kernel, re-played on the registers of the row's 16 threads: there G
  // lanes (G = sample_group(A) <= 32) hold one sample, lane j sums terms j, j + G, ... in turn,
  // then a xor tree runs (lane j adds lane j ^ off).  Thread `slot` holds the terms of lanes slot
  // and slot + 16 (and what the first loop adds to them): the tree's level 16 is one addition in
  // the thread, the levels below are shuffles among the row's 16 lanes — the same additions with
  // the same operands in the same order, no LDS, no second barrier.
  const int G = sample_group(A);
  float lo = term4[0], hi = term4[1];
  if (slot + 32 < A) lo = lo + term4[2];              // (first loop: v[aa % 32] += v[aa], A > 32)
  if (slot + 48 < A) hi = hi + term4[3];
  float val = G == 32 ? lo + hi : lo;                 // (level 16)
  for (int off = min(G, 16) >> 1; off >= 1; off >>= 1) val = val + __shfl_xor(val, off, 16);
  if (slot == 0 && ok) post_logp[grow] = val;
}

// The pass of workgroup (bx, net): 16 batch rows of one network.  A kernel of its own
// (mlp_forward_kernel) or one stage of q_chain_kernel.
__device__ __forceinline__ void mlp_forward_body(const MlpFwdArgs& a, const int net, const int bx,
                                                 float* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: uniform branches
  const int m = lane & 15, kg = lane >> 4;            // A operand row / D column; k group
  const int H = a.H, pitch = H + 4, tiles = H / 16;
  const int r0 = bx * kRows;
  // developer probe: wall-clock stamps (10 ns ticks) of workgroup (0, 0) at the phase boundaries
  const bool probe = a.stamps != nullptr && bx == 0 && net == 0 && tid == 0;
  auto stamp = [&](int i) {
    if (probe) {
      __builtin_amdgcn_sched_barrier(0);
      a.stamps[i] = wall_clock64();
      a.stamps[8 + i] = __builtin_readcyclecounter();        // shader clock (s_memtime)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  const int row = min(r0 + m, a.B - 1);               // batch row of this lane (clamped)
  const bool row_ok = r0 + m < a.B;
  const bool second = net >= a.split;                 // scalar
  // what follows the heads: the first network's tail, or (second parameter set / input) tail2 —
  // selects on a scalar condition, not indexing (a runtime index into the kernel arguments would
  // go to scratch)
  const int post = second ? a.tail2.post : a.post;
  const int64_t poff = net * a.stride_params + (second ? a.second_params : 0);
  const float* W1 = a.W1 + poff;
  const float* b1 = a.b1 + poff;
  const float* W2 = a.W2 + poff;
  const float* b2 = a.b2 + poff;
  float* h1g = a.h1 + net * a.stride_hidden;
  float* h2g = a.h2 + net * a.stride_hidden;
  float* hx = lds;
  float* hy = lds + kRows * pitch;
  if (a.reset_area != nullptr) {
    // the launch AHEAD of the chained ones empties their exchange area (kExchangeEmpty everywhere):
    // plain stores, published by the kernel boundary
    const int row_tiles = (a.B + kRows - 1) / kRows;
    const int64_t first = ((int64_t)net * (a.reset_blocks > 0 ? row_tiles : (int)gridDim.x) + bx) * blockDim.x + tid;
    const int64_t stride = (int64_t)(a.reset_blocks > 0 ? a.reset_blocks : (int)(gridDim.x * gridDim.y)) * blockDim.x;
    unsigned* area = reinterpret_cast<unsigned*>(a.reset_area);
    for (int64_t i = first; i < a.reset_floats; i += stride) area[i] = kExchangeEmpty;
    if (first == 0 && a.reset_failed != nullptr) *a.reset_failed = 0u;   // (and their failure word)
  }
  // wave w owns tiles w, w + 4, w + 8, w + 12; an index beyond the layer is clamped (the wave
  // then recomputes the last tile and drops it: no branch around an MFMA)
  int tile_of[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) tile_of[j] = min(wave + 4 * j, tiles - 1);

  // everything that does not depend on activations is requested up front
  f32x4 bias1[kMaxTiles], bias2[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    const f32x4_dword q1 = *reinterpret_cast<const f32x4_dword*>(b1 + 16 * tile_of[j] + 4 * kg);
    const f32x4_dword q2 = *reinterpret_cast<const f32x4_dword*>(b2 + 16 * tile_of[j] + 4 * kg);
    bias1[j] = f32x4{q1[0], q1[1], q1[2], q1[3]};
    bias2[j] = f32x4{q2[0], q2[1], q2[2], q2[3]};
  }
  // head tiles: wave w forms rows [16 * tile, 16 * tile + 16) of head w / tiles_per_head
  const int tiles_per_head = (a.NH + 15) / 16;
  // A single output (the value of a critic) is no MFMA tile: it is folded into the epilogue of
  // the second layer as a dot product (below).
  const bool value_head = a.heads == 1 && a.NH == 1 && a.act[0] == ACT_NONE;       // scalar
  const bool head_wave = !value_head && wave < a.heads * tiles_per_head;
  const int head = head_wave ? wave / tiles_per_head : 0;
  const int head_tile = head_wave ? wave - head * tiles_per_head : 0;
  // (selects, not array indexing: a runtime index into the kernel arguments would go to scratch)
  const float* Wh = (head == 0 ? a.Wh[0] : a.Wh[1]) + poff;
  const float* bh = (head == 0 ? a.bh[0] : a.bh[1]) + poff;
  f32x4 wvalue[kMaxTiles];                            // value head: w3 of this lane's features
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j)
    wvalue[j] = value_head ? load_w4(Wh, 16 * tile_of[j] + 4 * kg) : f32x4{0.f, 0.f, 0.f, 0.f};
  float hbias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) hbias[e] = bh[min(16 * head_tile + 4 * kg + e, a.NH - 1)];

  // hidden layer epilogue: bias + ReLU, to HBM (for the backward) and to an LDS image [row][feature]
  auto finish = [&](f32x4 (&acc)[kMaxTiles], const f32x4 (&bias)[kMaxTiles], float* global,
                    float* image) {
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      if (wave + 4 * j >= tiles) break;               // scalar condition
      const int f = 16 * tile_of[j] + 4 * kg;         // D rows 4*kg + e of the tile, column m
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = fmaxf(acc[j][e] + bias[j][e], 0.f);
      *reinterpret_cast<f32x4*>(image + m * pitch + f) = h;
      if (row_ok) *reinterpret_cast<f32x4_dword*>(global + (int64_t)(r0 + m) * a.ldh + f) =
          f32x4_dword{h[0], h[1], h[2], h[3]};
    }
  };

  unsigned rows1[kMaxTiles], rows2[kMaxTiles];       // first element of this lane's weight rows
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    rows1[j] = (unsigned)(16 * tile_of[j] + m) * a.ldw1;
    rows2[j] = (unsigned)(16 * tile_of[j] + m) * a.ldw2;
  }
  f32x4 acc[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xrow = (second ? a.X2 : a.X) + (int64_t)row * a.ldx;
  // the full k-chunks of the input row as buffer loads as well (lane: row and k group, scalar: the chunk)
  const __amdgpu_buffer_rsrc_t xbuffer = weight_buffer(second ? a.X2 : a.X);
  const unsigned xlane = 4u * ((unsigned)row * (unsigned)a.ldx + 4u * kg);
  Layer<kMaxTiles> l1;
  l1.start(W1, rows1, a.K1, kg, a.ldw1);
  stamp(1);
  l1.run(acc, [&](int k) { return buffer4(xbuffer, xlane, 4u * (unsigned)(k - 4 * kg)); },
         [&](int k) { return load_k4_tail(xrow, k, a.K1); });
  stamp(2);
  Layer<kMaxTiles> l2;
  l2.start(W2, rows2, H, kg, a.ldw2);                 // W2's first operands fly over the epilogue
  finish(acc, bias1, h1g, hx);
  __syncthreads();
  stamp(3);

#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto from_hx = [&](int k) { return *reinterpret_cast<const f32x4*>(hx + m * pitch + k); };
  l2.run(acc, from_hx, from_hx);
  stamp(4);
  // A head tile is one dependent chain of H / 4 MFMAs on one wave with nothing to hide a load
  // behind: its whole weight row (H <= 256: 16 loads per lane) is requested here, over the
  // second layer's epilogue and barrier.
  constexpr int kHeadChunks = 4 * kMaxTiles;
  const float* head_row = Wh + (int64_t)min(16 * head_tile + m, a.NH - 1) * a.ldw2;
  f32x4 wh[kHeadChunks];
  if (head_wave) {
#pragma unroll
    for (int c = 0; c < kHeadChunks; ++c)
      wh[c] = load_w4(head_row, 16 * min(c, tiles - 1) + 4 * kg);
  }
  __builtin_amdgcn_sched_barrier(0);                  // (or the loads sink to their use)
  float* partial = lds + 2 * kRows * pitch;           // [4 waves][16 rows]
  if (value_head) {
    // q[row] = b3 + sum_f h2[row][f] * w3[f]: this lane's 16 features, then the four k groups
    // of the row (lanes m, m + 16, m + 32, m + 48), then the four waves through LDS
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      if (wave + 4 * j >= tiles) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) part += fmaxf(acc[j][e] + bias2[j][e], 0.f) * wvalue[j][e];
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (kg == 0) partial[16 * wave + m] = part;
  }
  finish(acc, bias2, h2g, hy);
  __syncthreads();
  stamp(5);
  if (value_head) {
    if (wave == 0 && kg == 0 && row_ok) {
      float* out_base = a.out[0];
      const float q = ((partial[m] + partial[16 + m]) + (partial[32 + m] + partial[48 + m])) + hbias[0];
      out_base[net * a.stride_out + (int64_t)(r0 + m) * a.ldo] = q;
      if (a.xq != nullptr) {       // read by other workgroups of the same launch (ValueLines)
        const int slot = net < a.split ? 32 * net : 32 * (2 + net - a.split);     // one line per writer
        exchange_write(a.xq + (int64_t)bx * kExchangeTileFloats + slot + m, q);
      }
    }
    stamp(6);
    return;
  }

  // heads: one [16 outputs][16 rows] tile per head wave
  if (head_wave) {
    f32x4 out[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < kHeadChunks; ++c) {
      if (c < tiles) {                                // scalar
        const f32x4 b = *reinterpret_cast<const f32x4*>(hy + m * pitch + 16 * c + 4 * kg);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[0] = mfma16(wh[c][e], b[e], out[0]);
      }
    }
    float* out_base = head == 0 ? a.out[0] : a.out[1];
    const int act = head == 0 ? a.act[0] : a.act[1];
    float* dst = out_base + net * a.stride_out + (int64_t)(r0 + m) * a.ldo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = 16 * head_tile + 4 * kg + e;
      if (o < a.NH) {
        float v = out[0][e] + hbias[e];
        if (act == ACT_TANH) v = tanhf(v);
        if (row_ok) dst[o] = v;
        if (post != POST_NONE) lds[a.tail_offset + (head * kRows + m) * kPostPitch + o] = v;   // (H >= 188: hx, free now)
      }
    }
  }
  stamp(6);
  if (post == POST_NONE) return;                    // scalar

  policy_tail(a, second, r0, lds);
  stamp(7);
}


// Fused input-gradient chain of the same networks (the backward of mlp_forward_kernel without the
// weight gradients, which contract over the batch and go out as one grouped GEMM launch):
//   dz2 = dOut . W_out   * relu'(h2)      critic: dOut = dq [B], W_out = w3 [H]   (outer product)
//                                          actor : dOut = d head(s) [B, ldh], W_out = Wh [NH, H]
//   dz1 = dz2 . W2       * relu'(h1)
//   dxa = (dz1 . W1)[:, first : first + count]      optional: the action columns of the critic's
//                                                    input gradient, all the actor step needs
// Same decomposition as the forward: 16 batch rows per workgroup, wave w owns feature tiles w,
// w+4, ..., D[feature][row] products, LDS exchange between the layers; the weights are walked
// along their rows (W^T), i.e. k-strided loads.  dz2 / dz1 are written to HBM for the weight
// gradients.
// Chained launches: what other workgroups of the SAME launch wrote (l_tq, l_q, hb_dxa*) is read
// with exchange_read, and dxa is written with exchange_write (mlpfwd.h).
__device__ __forceinline__ float load_shared(const float* p, unsigned* exchange_failed) {
  return exchange_failed != nullptr ? exchange_read(p, exchange_failed) : *p;
}

// The logged sums of the step's loss over the whole batch (one workgroup; result -> l_stats)
__device__ __forceinline__ void mlp_loss_stats(const MlpBwdArgs& a) {
  const int tid = threadIdx.x;
  unsigned* co = a.exchange_failed;                   // null: plain loads
  double s0 = 0, s1 = 0, s2 = 0;
  for (int r = tid; r < a.B; r += blockDim.x) {
    if (a.loss == LOSS_TD) {
      const float y = td_target(a.l_rewards, a.l_discounts, a.l_tq, a.l_logp, a.l_alpha, r, a.l_tq_at,
                                a.l_nets, co);
      const float q1 = load_shared(a.l_q + a.l_q_at.index(0, r), co);
      const float e1 = q1 - y;
      float sq = e1 * e1;
      s1 += q1;
      if (a.l_nets == 2) {
        const float q2 = load_shared(a.l_q + a.l_q_at.index(1, r), co);
        const float e2 = q2 - y;
        sq = sq + e2 * e2;
        s2 += q2;
      }
      s0 += sq;
    } else if (a.l_nets == 2) {
      s0 += a.l_alpha * a.l_logp[r] -
            fminf(load_shared(a.l_q + a.l_q_at.index(0, r), co),
                  load_shared(a.l_q + a.l_q_at.index(1, r), co));
    } else {
      s0 += -load_shared(a.l_q + a.l_q_at.index(0, r), co);
    }
  }
  block_sum3(s0, s1, s2);
  if (tid == 0) {
    a.l_stats[0] = (float)s0; a.l_stats[1] = (float)s1; a.l_stats[2] = (float)s2;
    a.l_stats[3] = 0.f; a.l_stats[4] = 0.f; a.l_stats[5] = (float)a.B; a.l_stats[6] = 0.f;
    a.l_stats[7] = 0.f;
  }
}

// Chained launches (a.exchange_failed != null): what this pass consumes of its peers — the TD
// target's q, the twin's q, the critics' action-column gradients — is read (exchange_read) AFTER
// everything that does not depend on them (ReLU masks, the first weight operands, the head
// backward's own operands) has been requested: the hand-over's latency runs under those loads.
__device__ __forceinline__ void mlp_backward_body(const MlpBwdArgs& a, const int net, const int bx,
                                                  float* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int H = a.H, pitch = H + 4, tiles = H / 16;
  const int r0 = bx * kRows;
  unsigned* co = a.exchange_failed;                   // scalar; null: plain loads / stores
  const int row = min(r0 + m, a.B - 1);
  const bool row_ok = r0 + m < a.B;
  const float* W2 = a.W2 + net * a.stride_params;
  const float* h1g = a.h1 + net * a.stride_hidden;
  const float* h2g = a.h2 + net * a.stride_hidden;
  float* dz2g = a.dz2 + net * a.stride_hidden;
  float* dz1g = a.dz1 + net * a.stride_hidden;
  float* hx = lds;
  float* hy = lds + kRows * pitch;
  int tile_of[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) tile_of[j] = min(wave + 4 * j, tiles - 1);

  // The folded head backward, part 1 (first thing in the kernel): its operands are requested
  // through clamped addresses, no branch around a load, so that they return before the mask /
  // weight loads requested below — which then fly during the arithmetic of part 2.
  const bool formed = a.heads > 0 && a.hb_dxa0 != nullptr;      // scalar
  float* dhl = lds + 2 * kRows * pitch;             // [2 heads][16 rows][kHeadPitch] (formed only)
  constexpr int kHeadSlots = kPostPitch / 16;
  float hb_da[kHeadSlots], hb_second[kHeadSlots], hb_t[kHeadSlots], hb_sg[kHeadSlots],
      hb_ep[kHeadSlots], hb_pre[kHeadSlots];
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
  // ReLU masks of both layers (forward activations of this lane's rows / features), up front
  // H = 256: the dz1 product runs on INTERLEAVED tiles — wave w owns features [64 w, 64 w + 64),
  // tile j of them the features 64 w + 4 i + j — because W2^T is contiguous along the OUTPUT
  // index there: one 16-byte load per lane and k feeds all four tiles (4 loads per 16-k chunk
  // instead of 16 dword loads; the walk is bound by load instructions, ~6 ns of the CU's memory
  // pipe each).  In that layout register e of tile j is feature 64 w + 16 kg + 4 e + j.
  const bool wide = tiles == 4 * kMaxTiles;           // scalar
  f32x4 mask2[kMaxTiles], mask1[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) {
    const int f = 16 * tile_of[j] + 4 * kg;
    mask2[j] = load_k4(h2g + (int64_t)row * a.ldhid, f);
    mask1[j] = load_k4(h1g + (int64_t)row * a.ldhid, wide ? 64 * wave + 16 * kg + 4 * j : f);
  }
  // masked gradient of a hidden layer: to HBM (weight gradients) and to an LDS image [row][feature]
  auto finish = [&](const f32x4 (&acc)[kMaxTiles], const f32x4 (&mask)[kMaxTiles], float* global,
                    float* image) {
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      if (wave + 4 * j >= tiles) break;
      const int f = 16 * tile_of[j] + 4 * kg;
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = mask[j][e] > 0.f ? acc[j][e] : 0.f;
      *reinterpret_cast<f32x4*>(image + m * pitch + f) = d;
      if (row_ok) *reinterpret_cast<f32x4_dword*>(global + (int64_t)(r0 + m) * a.ldhid + f) =
          f32x4_dword{d[0], d[1], d[2], d[3]};
    }
  };

  unsigned cols2[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) cols2[j] = 16 * tile_of[j] + m;
  Layer<kMaxTiles, true> l1;                          // dz1 = W2^T dz2: requested before dz2 exists
  // wide: this lane's four output columns of row 4 kg (+ e) of a chunk — scalar row base + one
  // 32-bit lane offset (see Layer)
  const unsigned wcol = 4u * ((unsigned)(4 * kg) * a.ldw2 + 64 * wave + 4 * m);       // bytes
  const __amdgpu_buffer_rsrc_t w2_buffer = weight_buffer(W2);
  f32x4 wa[kHalf][4], wb[kHalf][4];                   // wide operand sets: [chunk][e] -> tiles 0..3
  auto wfill = [&](f32x4 (&w)[kHalf][4], int first) {
#pragma unroll
    for (int q = 0; q < kHalf; ++q) {
      const int c = min(first + q, tiles - 1);                    // past the end: re-read, unused
#pragma unroll
      for (int e = 0; e < 4; ++e)
        w[q][e] = buffer4(w2_buffer, wcol, 4u * (unsigned)((16 * c + e) * a.ldw2));
    }
  };
  if (wide) wfill(wa, 0);
  else l1.start(W2, cols2, H, kg, a.ldw2);

  if (co != nullptr) {                                // scalar
    if (formed) {
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
  }
  if (formed && a.hb_dxa1 != nullptr) {
#pragma unroll
    for (int u = 0; u < kHeadSlots; ++u) hb_da[u] = hb_da[u] + hb_second[u];
  }
  f32x4 acc[kMaxTiles];
#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.heads == 0) {                                 // critic: dq[row] * w3[feature]
    float dq;
    if (a.loss == LOSS_GIVEN) {
      dq = a.dq[net * a.stride_dq + row];
    } else {                                          // the step's loss, folded into this launch
      if (a.loss == LOSS_TD) {
        const float y = td_target(a.l_rewards, a.l_discounts, a.l_tq, a.l_logp, a.l_alpha, row,
                                  a.l_tq_at, a.l_nets, co);
        dq = 2.f * (load_shared(a.l_q + a.l_q_at.index(net, row), co) - y);
      } else {
        dq = actor_dq(a.l_q, row, a.l_q_at, a.l_nets == 2, net, co);
      }
      if (row_ok && wave == 0 && kg == 0)             // for the weight-gradient GEMM (dw3, db3)
        const_cast<float*>(a.dq)[net * a.stride_dq + row] = dq;
    }
    const float* w3 = a.w3 + net * a.stride_params;
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
      const f32x4 w = load_w4(w3, 16 * tile_of[j] + 4 * kg);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] = dq * w[e];
    }
  } else {                                            // actor: sum over the heads of dhead . Wh
    if (formed) {
      // part 2: actor_head_backward_kernel for the 16 rows of this workgroup, thread = (row, action
      // slot): the same expressions, so the same bits as the stand-alone launch; -> LDS (B operand
      // of the head products below) and -> dhead[.] in HBM (the weight-gradient GEMM contracts them)
      const int A = a.NH;
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
    for (int h = 0; h < a.heads; ++h) {
      const float* Wh = (h == 0 ? a.Wh[0] : a.Wh[1]) + net * a.stride_params;
      const float* dh = (h == 0 ? a.dhead[0] : a.dhead[1]) + (int64_t)row * a.ldh;
      const float* dl = dhl + (h * kRows + m) * kHeadPitch;
      Layer<kMaxTiles, true> lh;
      lh.start(Wh, cols2, a.NH, kg, a.ldw2);
      if (formed) {
        auto from_dl = [&](int k) { return *reinterpret_cast<const f32x4*>(dl + k); };
        lh.run(acc, from_dl, from_dl);
      } else {
        auto from_dh = [&](int k) { return load_k4(dh, k); };     // rows are padded to ldh >= 16
        lh.run(acc, from_dh, from_dh);
      }
    }
  }
  finish(acc, mask2, dz2g, hx);
  __syncthreads();

#pragma unroll
  for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto from_hx = [&](int k) { return *reinterpret_cast<const f32x4*>(hx + m * pitch + k); };
  if (wide) {
    f32x4 bA[kHalf], bB[kHalf];
    auto bfill = [&](f32x4 (&b)[kHalf], int first) {
#pragma unroll
      for (int q = 0; q < kHalf; ++q) b[q] = from_hx(16 * min(first + q, tiles - 1) + 4 * kg);
    };
    auto compute = [&](const f32x4 (&w)[kHalf][4], const f32x4 (&b)[kHalf]) {
#pragma unroll
      for (int q = 0; q < kHalf; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int j = 0; j < kMaxTiles; ++j) acc[j] = mfma16(w[q][e][j], b[q][e], acc[j]);
        }
      }
    };
    bfill(bA, 0);
    for (int c = 0; c < tiles; c += 2 * kHalf) {      // 16 chunks: four rounds of two sets
      wfill(wb, c + kHalf); bfill(bB, c + kHalf);
      __builtin_amdgcn_sched_barrier(0);
      compute(wa, bA);
      __builtin_amdgcn_sched_barrier(0);
      wfill(wa, c + 2 * kHalf); bfill(bA, c + 2 * kHalf);
      __builtin_amdgcn_sched_barrier(0);
      compute(wb, bB);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    l1.run(acc, from_hx, from_hx);
  }
  Layer<1, true> lx;
  const bool xa_wave = 16 * wave < a.xa_count;      // one 16-column tile per wave
  const unsigned colsx[1] = {(unsigned)(a.xa_first + min(16 * wave + m, max(a.xa_count - 1, 0)))};
  if (xa_wave) lx.start(a.W1 + net * a.stride_params, colsx, H, kg, a.ldw1);
  if (wide) {                                         // register e of tile j: feature .. + 4 e + j
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = 64 * wave + 16 * kg + 4 * e;
      f32x4 d;
#pragma unroll
      for (int j = 0; j < kMaxTiles; ++j) d[j] = mask1[e][j] > 0.f ? acc[j][e] : 0.f;
      *reinterpret_cast<f32x4*>(hy + m * pitch + f) = d;
      if (row_ok) *reinterpret_cast<f32x4_dword*>(dz1g + (int64_t)(r0 + m) * a.ldhid + f) =
          f32x4_dword{d[0], d[1], d[2], d[3]};
    }
  } else {
    finish(acc, mask1, dz1g, hy);
  }
  __syncthreads();

  if (xa_wave) {                                      // [16 action columns][16 rows] of dz1 . W1
    f32x4 out[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    auto from_hy = [&](int k) { return *reinterpret_cast<const f32x4*>(hy + m * pitch + k); };
    lx.run(out, from_hy, from_hy);
    if (row_ok) {
      float* dst = a.dxa + net * a.stride_dxa + (int64_t)(r0 + m) * a.ldxa;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = 16 * wave + 4 * kg + e;
        if (o < a.xa_count) {
          if (co != nullptr) exchange_write(dst + o, out[0][e]);
          else dst[o] = out[0][e];
        }
      }
    }
  }
  // the logged sums (uniform); a chain launch leaves them to its last workgroup
  if (a.loss != LOSS_GIVEN && co == nullptr && bx == 0 && net == 0) mlp_loss_stats(a);
}


#include "mlpimg_body.h"

// IMG: the passes on fp16x2 terms from weight images (mlpimg_body.h); else the float32 passes above.
template <bool IMG>
__global__ __launch_bounds__(256) void mlp_forward_kernel(MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];     // float32: two [16][H + 4] images; IMG: img_lds
  kernarg_prefetch<sizeof(MlpFwdArgs)>();
  if (a.store_on != 0 && blockIdx.x == gridDim.x - 1) {           // scalar: the store role of a collector step
    buffer_store_body(a.store, lds, a.lds_floats, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // its reads of the block have returned
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(a.done_flags + blockIdx.x, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if constexpr (IMG) mlp_forward_body_img(a, blockIdx.y, blockIdx.x, lds);
  else mlp_forward_body(a, blockIdx.y, blockIdx.x, lds);
  if (a.done_flags != nullptr) {                      // scalar: a step of a collector's block (see MlpFwdArgs)
    if (!IMG && a.rows_out != nullptr) {              // (the image pass wrote them from its input registers)
      const int r0 = blockIdx.x * kRows;
      for (int i = threadIdx.x; i < kRows * a.K1; i += blockDim.x) {
        const int r = i / a.K1, k = i - r * a.K1;
        if (r0 + r < a.B) a.rows_out[(int64_t)(r0 + r) * a.rows_ld + k] = a.X[(int64_t)(r0 + r) * a.ldx + k];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // this wave's stores (the actions) -> the system
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(a.done_flags + blockIdx.x, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool IMG>
__global__ __launch_bounds__(256) void mlp_backward_kernel(MlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  kernarg_prefetch<sizeof(MlpBwdArgs)>();
  if constexpr (IMG) mlp_backward_body_img(a, blockIdx.y, blockIdx.x, lds, 0);
  else mlp_backward_body(a, blockIdx.y, blockIdx.x, lds);
}

// ------------------------------------------------------------------ chained passes (mlpfwd.h)
// The launch's LAST workgroup is nobody's tile: it forms the logged sums of the step's loss from the
// exchanged values (waiting for each as it goes) while the other workgroups run their chains — in a
// launch of their own they were the by-product of workgroup (0, 0); at the end of the last chain
// they would be 2 - 3 us of the critical path.
__device__ __forceinline__ void chain_stats_role(const MlpBwdArgs& b) {
  mlp_loss_stats(b);                                  // (waits for every value it reads)
  if (threadIdx.x == 0 &&
      __hip_atomic_load(b.exchange_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    b.l_stats[0] = __builtin_nanf("");                // a value that never came: not a training step
  }                                                   // (the word stays: the optimizer epilogues of this
                                                      //  iteration skip their step, AdamFold::skip; the
                                                      //  next iteration's first launch clears it)
}

template <bool IMG>
__global__ __launch_bounds__(256) void q_critic_step_kernel(QCriticStep c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  kernarg_prefetch<sizeof(QCriticStep)>();
  const int roles = 2 * c.nets;
  const int tile = blockIdx.x / roles, role = blockIdx.x - tile * roles;      // scalar
  const int tiles = (c.fwd.B + kRows - 1) / kRows;
  const int own = tiles * roles;                      // scalar
  if ((int)blockIdx.x == own) {                       // the extra workgroup: the logged sums
    chain_stats_role(c.bwd);
    return;
  }
  if ((int)blockIdx.x > own) {                        // behind it: the NEXT iteration's policy passes (QCriticStep)
    if constexpr (IMG) {
      const int j = (int)blockIdx.x - own - 1;
      mlp_forward_body_img(c.ahead, j / tiles, j - (j / tiles) * tiles, lds);
    }
    return;
  }
  if (c.lose_first_target != 0 && blockIdx.x == 0) return;    // test hook: a workgroup that never answers
  if constexpr (IMG) mlp_forward_body_img(c.fwd, role, tile, lds);
  else mlp_forward_body(c.fwd, role, tile, lds);      // (its values -> the tile's exchange lines)
  if (role < c.nets) return;                          // a target
  __syncthreads();                                    // (the forward's LDS images are free)
  if constexpr (IMG) mlp_backward_body_img(c.bwd, role - c.nets, tile, lds, c.fwd.K1);
  else mlp_backward_body(c.bwd, role - c.nets, tile, lds);
}

template <bool IMG>
__global__ __launch_bounds__(256) void q_actor_step_kernel(QActorStep c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  kernarg_prefetch<sizeof(QActorStep)>();
  const int roles = c.used + 1;
  const int tile = blockIdx.x / roles, role = blockIdx.x - tile * roles;      // scalar
  const int tiles = (c.fwd.B + kRows - 1) / kRows;
  if (tile == tiles) {
    chain_stats_role(c.bwd);
    return;
  }
  if (role == c.used) {                               // the actor: both critics' action columns
    if constexpr (IMG) mlp_backward_body_img(c.actor, 0, tile, lds, c.fwd.K1);
    else mlp_backward_body(c.actor, 0, tile, lds);
    return;
  }
  if constexpr (IMG) {
    mlp_forward_body_img(c.fwd, role, tile, lds);
    __syncthreads();
    mlp_backward_body_img(c.bwd, role, tile, lds, c.fwd.K1);
  } else {
    mlp_forward_body(c.fwd, role, tile, lds);
    __syncthreads();
    mlp_backward_body(c.bwd, role, tile, lds);        // (the twin's q; dxa -> the exchange area)
  }
}

}  // namespace

std::atomic<int> g_policy_tail{1};
std::atomic<int> g_q_chain{1};
std::atomic<int> g_q_images{1};
std::atomic<int> g_chain_fault{0};     // tuning key "chain_fault": the NEXT critic step loses its first workgroup (test hook)

// The tail's three [16][kPostPitch] images live in the first hidden image where they fit (H >= 188:
// the second image is still being read by other head waves), else behind both images.
bool mlp_policy_tail_supported(int H, int NH) { return H >= 16 && NH <= kPostPitch; }

static int policy_tail_offset(int H) {
  return 3 * kRows * kPostPitch <= kRows * (H + 4) ? 0 : 2 * kRows * (H + 4) + 4 * kRows;
}

bool mlp_forward_supported(int H, int NH, int heads) {
  return H >= 16 && H <= 16 * 4 * kMaxTiles && H % 16 == 0 && NH >= 1 && heads >= 1 && heads <= 2 &&
         heads * ((NH + 15) / 16) <= 4;
}

std::atomic<unsigned long long*> g_forward_stamps{nullptr};      // (gemm16.hip stamps its weight-gradient launches behind the forward ring)
static std::atomic<unsigned> g_forward_launches{0};

extern "C" int tonic_debug_forward_stamps(uint64_t* d_stamps) {
  g_forward_stamps.store(reinterpret_cast<unsigned long long*>(d_stamps));
  g_forward_launches.store(0);
  return TONIC_OK;
}

// The image passes' LDS (img_lds) can exceed the 64 KB a kernel gets by default: raised once per kernel.
static bool image_pass_supported(int K1, int H) {
  return H >= 16 && H <= 16 * 4 * kMaxTiles && H % 16 == 0 && K1 >= 0 && img_lds(K1, H).total <= kImgLdsCap;
}
bool mlp_image_pass_supported(int K1, int H) { return image_pass_supported(K1, H); }

template <typename Kernel>
static int allow_image_lds(Kernel kernel, const char* what) {
  static thread_local std::vector<const void*> configured;
  const void* fn = reinterpret_cast<const void*>(kernel);
  for (const void* known : configured) if (known == fn) return TONIC_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kImgLdsCap);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%d B LDS): %s", what, kImgLdsCap, hipGetErrorString(e));
    return TONIC_ERR_LAUNCH;
  }
  configured.push_back(fn);
  return TONIC_OK;
}

int launch_mlp_forward(const MlpFwdArgs& a, int nets, hipStream_t stream) {
  TONIC_REQUIRE(mlp_forward_supported(a.H, a.NH, a.heads) && a.B > 0 && a.K1 > 0 && nets > 0,
                TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: H=%d NH=%d heads=%d B=%d K1=%d", a.H,
                a.NH, a.heads, a.B, a.K1);
  TONIC_REQUIRE(a.ldw1 >= a.K1 && a.ldw1 % 4 == 0 && a.ldw2 >= a.H && a.ldw2 % 4 == 0 &&
                    a.ldh >= a.H && a.ldh % 4 == 0,
                TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: strides %d / %d / %d", a.ldw1, a.ldw2, a.ldh);
  TONIC_REQUIRE(a.enc_out == nullptr ||
                    (a.post != POST_NONE && a.enc_obs && a.enc_mean && a.enc_std && a.enc_O > 0 &&
                     a.enc_ld >= a.enc_O + a.NH && (a.enc_out2 == nullptr || (a.enc_obs2 && a.enc_act2))),
                TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: encoder in the policy tail (O=%d ld=%d)",
                a.enc_O, a.enc_ld);
  TONIC_REQUIRE(a.tail2.post == POST_NONE ||
                    (nets == 2 && a.split == 1 && a.post != POST_NONE && a.tail2.actions != nullptr &&
                     (a.tail2.enc_out == nullptr || (a.tail2.enc_obs && a.enc_mean && a.enc_std)) &&
                     (a.tail2.post == POST_SQUASHED_SAMPLE ? a.heads == 2
                                                           : a.heads == 1 && (a.tail2.post == POST_COPY || a.tail2.eps))),
                TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: second policy tail %d (nets=%d split=%d)",
                a.tail2.post, nets, a.split);
  TONIC_REQUIRE(a.post == POST_NONE ||
                    ((nets == 1 || a.tail2.post != POST_NONE) && a.NH <= kPostPitch && a.post_actions != nullptr &&
                     mlp_policy_tail_supported(a.H, a.NH) &&
                     (a.post == POST_SQUASHED_SAMPLE ? a.heads == 2
                                                     : a.heads == 1 && (a.post == POST_COPY || a.post_eps))),
                TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: policy tail %d with heads=%d NH=%d H=%d",
                a.post, a.heads, a.NH, a.H);
  TONIC_REQUIRE(a.split >= nets || a.X2 != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "mlp_forward: split=%d of %d networks without a second input", a.split, nets);
  size_t lds = (2 * (size_t)kRows * (a.H + 4) + 4 * kRows) * sizeof(float);
  MlpFwdArgs launch = a;
  launch.tail_offset = 0;
  if (a.post != POST_NONE) {
    launch.tail_offset = policy_tail_offset(a.H);
    const size_t need = ((size_t)launch.tail_offset + 3 * kRows * kPostPitch) * sizeof(float);
    if (need > lds) lds = need;
  }
  launch.stamps = nullptr;
  if (unsigned long long* base = g_forward_stamps.load()) {      // developer probe: ring of 8 launches
    launch.stamps = base + 16 * (g_forward_launches.fetch_add(1) % 8);
  }
  if (a.img.block != nullptr) {
    TONIC_REQUIRE(image_pass_supported(a.K1, a.H), TONIC_ERR_INVALID_ARGUMENT,
                  "mlp_forward: K1=%d H=%d outside the image pass", a.K1, a.H);
    const ImgLds L = img_lds(a.K1, a.H);
    launch.tail_offset = L.off_f32 / 4;
    launch.lds_floats = L.total / 4;
    TONIC_REQUIRE(a.store_on == 0 || (nets == 1 && a.done_flags != nullptr && a.store.O <= L.total / 4),
                  TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: a store role needs a single-network collector step");
    const int status = allow_image_lds(mlp_forward_kernel<true>, "mlp_forward_kernel");
    if (status != TONIC_OK) return status;
    hipLaunchKernelGGL(mlp_forward_kernel<true>, dim3((a.B + kRows - 1) / kRows + (a.store_on != 0 ? 1 : 0), nets),
                       dim3(256), L.total, stream, launch);
  } else {
    TONIC_REQUIRE(a.store_on == 0, TONIC_ERR_INVALID_ARGUMENT, "mlp_forward: the store role rides on the image pass");
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3((a.B + kRows - 1) / kRows, nets), dim3(256), lds,
                       stream, launch);
  }
  TONIC_CHECK_LAUNCH("mlp_forward_kernel");
  return TONIC_OK;
}

bool mlp_backward_supported(int H, int NH, int heads, int xa_count) {
  return H >= 16 && H <= 16 * 4 * kMaxTiles && H % 16 == 0 && heads >= 0 && heads <= 2 &&
         (heads == 0 || (NH >= 1 && NH <= 64)) && xa_count >= 0 && xa_count <= 64;
}

int launch_mlp_backward(const MlpBwdArgs& a, int nets, hipStream_t stream) {
  TONIC_REQUIRE(mlp_backward_supported(a.H, a.NH, a.heads, a.xa_count) && a.B > 0 && nets > 0,
                TONIC_ERR_INVALID_ARGUMENT, "mlp_backward: H=%d NH=%d heads=%d B=%d", a.H, a.NH,
                a.heads, a.B);
  TONIC_REQUIRE(a.ldw2 >= a.H && (a.xa_count == 0 || a.ldw1 >= a.K1) && a.ldhid >= a.H &&
                    a.ldhid % 4 == 0, TONIC_ERR_INVALID_ARGUMENT,
                "mlp_backward: strides %d / %d / %d", a.ldw1, a.ldw2, a.ldhid);
  TONIC_REQUIRE(a.hb_dxa0 == nullptr ||
                    (a.heads >= 1 && nets == 1 && a.NH <= kPostPitch && a.hb_act && a.dhead[0] &&
                     (!a.hb_sac || (a.heads == 2 && a.hb_eps && a.hb_sigma && a.hb_spre && a.dhead[1]))),
                TONIC_ERR_INVALID_ARGUMENT, "mlp_backward: folded head backward (heads=%d NH=%d)",
                a.heads, a.NH);
  const size_t lds = (2 * (size_t)kRows * (a.H + 4) +
                      (a.hb_dxa0 != nullptr ? 2 * (size_t)kRows * kHeadPitch : 0)) * sizeof(float);
  if (a.img.block != nullptr) {
    TONIC_REQUIRE(image_pass_supported(0, a.H), TONIC_ERR_INVALID_ARGUMENT,
                  "mlp_backward: H=%d outside the image pass", a.H);
    const int status = allow_image_lds(mlp_backward_kernel<true>, "mlp_backward_kernel");
    if (status != TONIC_OK) return status;
    hipLaunchKernelGGL(mlp_backward_kernel<true>, dim3((a.B + kRows - 1) / kRows, nets), dim3(256),
                       img_lds(0, a.H).total, stream, a);
  } else {
    hipLaunchKernelGGL(mlp_backward_kernel<false>, dim3((a.B + kRows - 1) / kRows, nets), dim3(256), lds,
                       stream, a);
  }
  TONIC_CHECK_LAUNCH("mlp_backward_kernel");
  return TONIC_OK;
}

static size_t chain_lds_bytes(int H) {
  return (2 * (size_t)kRows * (H + 4) + 4 * kRows + 2 * (size_t)kRows * kHeadPitch) * sizeof(float);
}

int launch_q_critic_step(const QCriticStep& c, hipStream_t stream) {
  const MlpFwdArgs& f = c.fwd;
  TONIC_REQUIRE(c.nets >= 1 && c.nets <= 2 && c.bwd.exchange_failed != nullptr && f.split == c.nets &&
                    f.post == POST_NONE && f.heads == 1 && f.NH == 1 && f.xq != nullptr &&
                    mlp_forward_supported(f.H, 1, 1) && c.bwd.heads == 0 && c.bwd.loss == LOSS_TD &&
                    c.bwd.B == f.B && c.bwd.H == f.H &&
                    mlp_backward_supported(f.H, 1, 0, c.bwd.xa_count),
                TONIC_ERR_INVALID_ARGUMENT, "q_critic_step: nets=%d H=%d", c.nets, f.H);
  const int tiles = (f.B + kRows - 1) / kRows;
  QCriticStep launch = c;
  launch.lose_first_target = g_chain_fault.exchange(0);      // (one launch, then off again)
  const bool images = f.img.block != nullptr;
  TONIC_REQUIRE(images == (c.bwd.img.block != nullptr) && (!images || image_pass_supported(f.K1, f.H)),
                TONIC_ERR_INVALID_ARGUMENT, "q_critic_step: weight images for one half only (K1=%d H=%d)", f.K1, f.H);
  TONIC_REQUIRE(c.ahead_nets == 0 ||
                    (images && c.ahead_nets <= 2 && c.ahead.img.block != nullptr && c.ahead.B == f.B &&
                     c.ahead.H == f.H && c.ahead.K1 <= f.K1 && c.ahead.post != POST_NONE &&
                     c.ahead.done_flags == nullptr && c.ahead.store_on == 0 &&
                     (c.ahead_nets == 1 || (c.ahead.split == 1 && c.ahead.tail2.post != POST_NONE))),
                TONIC_ERR_INVALID_ARGUMENT, "q_critic_step: %d policy passes ahead (images %d)", c.ahead_nets,
                (int)images);
  if (c.ahead_nets > 0) {                               // (what launch_mlp_forward sets for a launch of its own)
    const ImgLds L = img_lds(c.ahead.K1, c.ahead.H);
    launch.ahead.tail_offset = L.off_f32 / 4;
    launch.ahead.lds_floats = L.total / 4;
    launch.ahead.stamps = nullptr;
    launch.ahead.reset_blocks = tiles * c.ahead_nets;
  }
  if (images) {
    const int status = allow_image_lds(q_critic_step_kernel<true>, "q_critic_step_kernel");
    if (status != TONIC_OK) return status;
    hipLaunchKernelGGL(q_critic_step_kernel<true>, dim3(tiles * 2 * c.nets + 1 + tiles * c.ahead_nets), dim3(256),
                       img_lds(f.K1, f.H).total, stream, launch);
  } else {
    hipLaunchKernelGGL(q_critic_step_kernel<false>, dim3(tiles * 2 * c.nets + 1), dim3(256),
                       chain_lds_bytes(f.H), stream, launch);
  }
  TONIC_CHECK_LAUNCH("q_critic_step_kernel");
  return TONIC_OK;
}

int launch_q_actor_step(const QActorStep& c, hipStream_t stream) {
  const MlpFwdArgs& f = c.fwd;
  TONIC_REQUIRE(c.used >= 1 && c.used <= 2 && c.bwd.exchange_failed != nullptr && f.post == POST_NONE &&
                    f.heads == 1 && f.NH == 1 && f.xq != nullptr && f.split >= c.used &&
                    mlp_forward_supported(f.H, 1, 1) && c.bwd.heads == 0 &&
                    c.bwd.loss == LOSS_ACTOR && c.bwd.B == f.B &&
                    c.bwd.H == f.H && mlp_backward_supported(f.H, 1, 0, c.bwd.xa_count) &&
                    c.actor.heads >= 1 && c.actor.hb_dxa0 != nullptr &&
                    c.actor.exchange_failed != nullptr &&
                    c.actor.B == f.B && c.actor.H == f.H && c.actor.NH <= kPostPitch &&
                    mlp_backward_supported(f.H, c.actor.NH, c.actor.heads, c.actor.xa_count),
                TONIC_ERR_INVALID_ARGUMENT, "q_actor_step: used=%d H=%d", c.used, f.H);
  const int tiles = (f.B + kRows - 1) / kRows;
  const bool images = f.img.block != nullptr;
  TONIC_REQUIRE(images == (c.bwd.img.block != nullptr) && images == (c.actor.img.block != nullptr) &&
                    (!images || image_pass_supported(f.K1, f.H)),
                TONIC_ERR_INVALID_ARGUMENT, "q_actor_step: weight images for some roles only (K1=%d H=%d)", f.K1, f.H);
  if (images) {
    const int status = allow_image_lds(q_actor_step_kernel<true>, "q_actor_step_kernel");
    if (status != TONIC_OK) return status;
    hipLaunchKernelGGL(q_actor_step_kernel<true>, dim3(tiles * (c.used + 1) + 1), dim3(256),
                       img_lds(f.K1, f.H).total, stream, c);
  } else {
    hipLaunchKernelGGL(q_actor_step_kernel<false>, dim3(tiles * (c.used + 1) + 1), dim3(256),
                       chain_lds_bytes(f.H), stream, c);
  }
  TONIC_CHECK_LAUNCH("q_actor_step_kernel");
  return TONIC_OK;
}

}  // namespace tonic
