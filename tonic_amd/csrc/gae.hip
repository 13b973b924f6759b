// Lambda-return / GAE scan + advantage statistics (HBM-bound), gfx950.
//
// Restates tonic/replays/utils.py:4-19 (reverse scan over T, independent per worker column)
// and tonic/replays/segments.py:41-46 (advantages = returns - values, global mean and
// population std).  Layout: every array is [T, W] row-major float32, so a time row is W
// contiguous floats and lane <-> worker gives perfectly coalesced row accesses.
//
// Parallelisation: one lane per worker column.  With ONE chunk a lane walks the whole T axis with
// the reference's float32 operation order (compiled with -ffp-contract=off): returns bit-identical
// to the reference.  With more chunks T is cut into 128- or 64-row segments that run concurrently and
// exchange affine carries (gae_onepass_kernel below): one pass over the data, ~1e-6 relative.
#include <atomic>
#include <type_traits>

#include "common.h"

namespace tonic {

constexpr int kGaeThreads = 256;
constexpr int kUnroll = 8;        // time rows loaded ahead of the dependent chain

struct GaeArgs {
  const float* next_values;
  const float* rewards;
  const float* resets;
  const float* terminations;
  const float* values;
  float* returns;
  float* advantages;
  float* carry_a;     // [segments, W]: the segments' affine maps (segmented pass only)
  float* carry_b;
  double* block_sums; // [blocks, 4] = {sum, sum_sq, min, max}
  int64_t T, W;
  float gamma, lambda, one_minus_lambda;
  int probe;          // developer probe of gae_stream_kernel (tuning key gae_stream = 2): helpers idle
};

// ---- more than one chunk: ONE pass over the data (28 B / transition) ------------------------------
// T is cut into segments of 128 (64) rows; a workgroup of 8 (4) waves owns one segment of one
// 64-column tile and keeps it ENTIRELY in registers (two to four workgroups per CU, so that one
// loads while another computes or stores): wave k holds rows [16k, 16k+16) of the segment
// (5 arrays x 16 rows per lane, all 80 loads in flight at once — that, not occupancy, is the
// memory-level parallelism).  Nothing is read twice:
//   1. every wave composes its 16 rows into an affine map ret_in -> a + b * ret_in;
//   2. the maps meet in LDS; wave 0 composes them into the segment's map and publishes it (release,
//      one flag per workgroup);
//   3. wave 0 forms the carry entering the segment by applying the maps of all LATER segments of its
//      column tile in order, starting from next_values[T-1] (8 B per lane and segment, fetched by
//      the waves side by side); later segments hold smaller tickets (below), so they are running
//      or done: no deadlock;
//   4. every wave applies the maps of the later waves of its own segment (LDS) to that carry and
//      replays the EXACT reference recurrence over its 16 rows from registers, writing returns and
//      raw advantages and the float64 moments.
// The summary / carry / scan triple this replaces read the four scan inputs twice (44 B).
constexpr int kSegRowsPerWave = 16;
// 8 waves (128-row segments, two workgroups per CU): fewer segments, shorter carry walks — the
// latency-bound sizes; 4 waves (64-row segments, four per CU): more workgroups in different phases
// per CU — the bandwidth-bound sizes (0.75 of the HBM peak at W = 65 536, against 0.70).
constexpr int64_t kWideColumns = 8192;
inline int seg_waves(int64_t W) { return W >= kWideColumns ? 4 : 8; }
constexpr int kFarSegments = 64;    // later segments whose maps a workgroup can hold in LDS

struct GaeOnePass {
  GaeArgs g;
  unsigned* ticket;     // zero at launch
  // carry_a / carry_b (the segment's map) and `inclusive` (the carry LEAVING the segment towards
  // earlier rows), [segments, W] each, are all-ones words (kCarryEmpty) at launch: every value is
  // written once with an agent-scope store and read with agent-scope loads until it is no longer
  // empty — the data is its own flag (mlpfwd.h, exchange_read: a flag word behind
  // `s_waitcnt vmcnt(0)`, round 2's protocol, can be seen before the data's write-through has
  // landed; a release fence in front of it costs 5x at W = 65 536).
  float* inclusive;
  int tiles, segments;
};

constexpr unsigned kCarryEmpty = 0xffffffffu;         // (hipMemsetAsync 0xFF; no carry is this NaN)
__device__ __forceinline__ unsigned carry_peek(const float* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float carry_wait(const float* p) {
  unsigned bits;
  while ((bits = carry_peek(p)) == kCarryEmpty) __builtin_amdgcn_s_sleep(2);
  return __uint_as_float(bits);
}

template <int kSegWaves>
__global__ __launch_bounds__(kSegWaves * 64, 4) void gae_onepass_kernel(GaeOnePass p) {
  constexpr int kSegRows = kSegWaves * kSegRowsPerWave;
  __shared__ float map_a[kSegWaves][64], map_b[kSegWaves][64], carry_in[64];
  __shared__ float far_a[kFarSegments][64], far_b[kFarSegments][64], far_incl[kSegWaves][64];
  __shared__ int far_state[kSegWaves], resolved;
  __shared__ double red[4][kSegWaves];
  __shared__ unsigned ticket;
  const GaeArgs& g = p.g;
  // Workgroups take their (segment, tile) in the order they START: later segments first.
  if (threadIdx.x == 0)
    ticket = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int seg = p.segments - 1 - (int)(ticket / (unsigned)p.tiles);
  const int tile = (int)(ticket % (unsigned)p.tiles);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t w = (int64_t)tile * 64 + lane;
  const bool column = w < g.W;
  const int64_t wc = column ? w : g.W - 1;
  const int64_t base = (int64_t)seg * kSegRows + wave * kSegRowsPerWave;   // first row of the wave
  constexpr int R = kSegRowsPerWave;
  float nv[R], r[R], rs[R], tm[R], v[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t t = min(base + u, g.T - 1);
    const int64_t i = t * g.W + wc;
    nv[u] = __builtin_nontemporal_load(g.next_values + i);
    r[u] = __builtin_nontemporal_load(g.rewards + i);
    rs[u] = __builtin_nontemporal_load(g.resets + i);
    tm[u] = __builtin_nontemporal_load(g.terminations + i);
    v[u] = __builtin_nontemporal_load(g.values + i);
  }
  // 1. the wave's rows as one affine map (rows past T are the identity)
  float ca = 0.f, cb = 1.f;
#pragma unroll
  for (int u = R - 1; u >= 0; --u) {
    if (base + u < g.T) {
      const float keep = 1.f - tm[u], cont = 1.f - rs[u];
      const float b = g.gamma * g.lambda * keep * cont;
      const float a = r[u] + g.gamma * keep * nv[u] * (cont * g.one_minus_lambda + rs[u]);
      ca = a + b * ca;
      cb = b * cb;
    }
  }
  map_a[wave][lane] = ca;
  map_b[wave][lane] = cb;
  __syncthreads();
  float seg_a = 0.f, seg_b = 1.f;
  if (wave == 0) {
    // 2. the segment's map, published for the EARLIER segments of this column tile
    float sa = 0.f, sb = 1.f;
#pragma unroll
    for (int k = kSegWaves - 1; k >= 0; --k) {
      sa = map_a[k][lane] + map_b[k][lane] * sa;
      sb = map_b[k][lane] * sb;
    }
    if (seg > 0 && column) {                // agent-scope stores: the values are their own flags
      __hip_atomic_store(g.carry_a + (int64_t)seg * g.W + w, sa, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.carry_b + (int64_t)seg * g.W + w, sb, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    seg_a = sa;
    seg_b = sb;
  }
  // 3. the carry entering this segment = the maps of the later segments applied, from the end of
  //    time towards this segment, to next_values[T-1] (utils.py:11).  A later segment that already
  //    knows its own carry has published the value it hands on (`inclusive`): the walk starts at
  //    the NEAREST such segment and only applies the maps in between — the same float32 operations
  //    as the full walk, so the result does not depend on timing.  The waves look at one later
  //    segment each per round, nearest first (flag, then 8-12 bytes per lane with agent-scope
  //    loads): a dependent walk costs a cross-XCD round trip per segment.
  const int later = p.segments - 1 - seg;
  int found = -1;
  for (int d0 = 0; d0 < later && found < 0; d0 += kSegWaves) {
    const int d = d0 + wave;
    if (d < later) {
      const int s2 = seg + 1 + d;
      // the LDS stash is full at distance kFarSegments - 1: that segment must hand over its carry
      const unsigned need = (d >= kFarSegments - 1 && d < later - 1) ? 2u : 1u;
      if (d < kFarSegments) {
        far_a[d][lane] = carry_wait(g.carry_a + (int64_t)s2 * g.W + wc);
        far_b[d][lane] = carry_wait(g.carry_b + (int64_t)s2 * g.W + wc);
      }
      // `inclusive`: taken when ALL columns of the tile have it (the walk below is the tile's, not
      // the lane's) — awaited only where the LDS stash ends
      const float* incl = p.inclusive + (int64_t)s2 * g.W + wc;
      unsigned bits = carry_peek(incl);
      if (need == 2u) bits = __float_as_uint(carry_wait(incl));
      const unsigned state = __all(bits != kCarryEmpty) ? 2u : 1u;
      if (state == 2u) far_incl[wave][lane] = __uint_as_float(bits);
      if (lane == 0) far_state[wave] = (int)state;
    } else if (lane == 0) {
      far_state[wave] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int hit = -1;
      for (int k = kSegWaves - 1; k >= 0; --k)
        if (far_state[k] == 2) hit = d0 + k;             // the nearest one wins
      resolved = hit;
    }
    __syncthreads();
    found = resolved;
    if (found < 0 && d0 + kSegWaves < later) __syncthreads();     // far_state is rewritten next round
  }
  if (wave == 0) {
    float carry;
    int from;
    if (found >= 0) {
      carry = far_incl[found % kSegWaves][lane];
      from = found - 1;
    } else {
      carry = g.next_values[(g.T - 1) * g.W + wc];
      from = later - 1;
    }
    for (int d = from; d >= 0; --d) carry = far_a[d][lane] + far_b[d][lane] * carry;
    carry_in[lane] = carry;
    if (seg > 0 && column)                // what this segment hands on to the earlier ones
      __hip_atomic_store(p.inclusive + (int64_t)seg * g.W + w, seg_a + seg_b * carry,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // 4. the later waves of this segment, then the exact recurrence over this wave's rows
  float last = carry_in[lane];
  for (int k = kSegWaves - 1; k > wave; --k) last = map_a[k][lane] + map_b[k][lane] * last;
  double sum = 0.0, sum_sq = 0.0;
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int u = R - 1; u >= 0; --u) {
    if (base + u < g.T) {
      float boot = g.one_minus_lambda * nv[u] + g.lambda * last;   // utils.py:13-14
      boot = boot * (1.f - rs[u]);                                 // :15
      boot = boot + rs[u] * nv[u];                                 // :16
      boot = boot * (1.f - tm[u]);                                 // :17
      last = r[u] + g.gamma * boot;                                // :18
      const float adv = last - v[u];                               // segments.py:42
      if (column) {
        const int64_t i = (base + u) * g.W + w;
        __builtin_nontemporal_store(last, g.returns + i);
        __builtin_nontemporal_store(adv, g.advantages + i);
        sum += (double)adv;
        sum_sq += (double)adv * (double)adv;
        lo = fminf(lo, adv);
        hi = fmaxf(hi, adv);
      }
    }
  }
  sum = wave_sum(sum);
  sum_sq = wave_sum(sum_sq);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = sum_sq; red[2][wave] = lo; red[3][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < kSegWaves; ++i) {
      s0 += red[0][i]; s1 += red[1][i];
      mn = fmin(mn, red[2][i]); mx = fmax(mx, red[3][i]);
    }
    const int64_t b = (int64_t)seg * p.tiles + tile;          // (not the ticket: fixed order)
    g.block_sums[4 * b] = s0;
    g.block_sums[4 * b + 1] = s1;
    g.block_sums[4 * b + 2] = mn;
    g.block_sums[4 * b + 3] = mx;
  }
}

// One chunk: the reference recurrence over the whole T axis, kUnroll rows of loads in flight.
__global__ __launch_bounds__(kGaeThreads) void gae_scan_kernel(GaeArgs g) {
  const int64_t w = (int64_t)blockIdx.x * kGaeThreads + threadIdx.x;
  const bool active = w < g.W;
  double sum = 0.0, sum_sq = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  if (active) {
    const int64_t t0 = 0, t1 = g.T;
    float last = g.next_values[(g.T - 1) * g.W + w];             // utils.py:11
    int64_t t = t1 - 1;
    for (; t - (kUnroll - 1) >= t0; t -= kUnroll) {
      float nv[kUnroll], r[kUnroll], rs[kUnroll], tm[kUnroll], v[kUnroll], ret[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (t - u) * g.W + w;
        nv[u] = __builtin_nontemporal_load(g.next_values + i);
        r[u] = __builtin_nontemporal_load(g.rewards + i);
        rs[u] = __builtin_nontemporal_load(g.resets + i);
        tm[u] = __builtin_nontemporal_load(g.terminations + i);
        v[u] = __builtin_nontemporal_load(g.values + i);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        float boot = g.one_minus_lambda * nv[u] + g.lambda * last;   // utils.py:13-14
        boot = boot * (1.f - rs[u]);                                 // :15
        boot = boot + rs[u] * nv[u];                                 // :16
        boot = boot * (1.f - tm[u]);                                 // :17
        last = r[u] + g.gamma * boot;                                // :18
        ret[u] = last;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = (t - u) * g.W + w;
        const float adv = ret[u] - v[u];                             // segments.py:42
        __builtin_nontemporal_store(ret[u], g.returns + i);
        __builtin_nontemporal_store(adv, g.advantages + i);
        sum += (double)adv;
        sum_sq += (double)adv * (double)adv;
        lo = fminf(lo, adv);
        hi = fmaxf(hi, adv);
      }
    }
    for (; t >= t0; --t) {
      const int64_t i = t * g.W + w;
      const float nv = g.next_values[i], rs = g.resets[i], tm = g.terminations[i];
      float boot = g.one_minus_lambda * nv + g.lambda * last;
      boot = boot * (1.f - rs);
      boot = boot + rs * nv;
      boot = boot * (1.f - tm);
      last = g.rewards[i] + g.gamma * boot;
      const float adv = last - g.values[i];
      g.returns[i] = last;
      g.advantages[i] = adv;
      sum += (double)adv;
      sum_sq += (double)adv * (double)adv;
      lo = fminf(lo, adv);
      hi = fmaxf(hi, adv);
    }
  }
  // block partial -> HBM (deterministic: fixed lane / wave order, no atomics)
  __shared__ double red[4][kGaeThreads / 64];
  sum = wave_sum(sum);
  sum_sq = wave_sum(sum_sq);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = sum_sq; red[2][wave] = lo; red[3][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < kGaeThreads / 64; ++i) {
      s0 += red[0][i]; s1 += red[1][i];
      mn = fmin(mn, red[2][i]); mx = fmax(mx, red[3][i]);
    }
    const int64_t b = blockIdx.x;
    g.block_sums[4 * b] = s0;
    g.block_sums[4 * b + 1] = s1;
    g.block_sums[4 * b + 2] = mn;
    g.block_sums[4 * b + 3] = mx;
  }
}

// ---- one chunk, few columns: the exact chain at the speed of its own dependent instructions ------
// The reference recurrence is 7 dependent float32 operations per time row and column.  A lane =
// column scan like gae_scan_kernel above needs W alone to fill the chip: at W = 256 (BASELINE cfg 2)
// four waves issue every load of the 29 MB and wait for each batch of 8 rows — 620 us whatever the
// size below W ~ 10 k, ten times the chain's own ~60 us.  Here a workgroup owns 16 columns over
// the WHOLE time axis (W / 16 workgroups: 16 at cfg 2, 80 at cfg 5's per-GPU share) and splits the
// work by role instead of by data:
//   * four helper waves stream the rows through LDS, 64 rows (one chunk) per barrier: they request
//     chunk i + 4 from HBM, turn chunk i + 2 — requested two barriers ago, so ~2 chunk times of
//     latency cover — into the operands the chain needs ((1 - lambda) nv, 1 - resets, resets nv,
//     1 - terminations: the reference's own intermediate roundings, utils.py:13-17) and write them
//     in the chain's read order, and store chunk i - 1's returns / advantages and fold their
//     float64 moments;
//   * ONE wave walks the chain: per row 7 dependent VALU instructions + 1.5 LDS instructions
//     (operands come as one ds_read_b128 per row + one per four rows, returns leave as one
//     ds_write_b128 per four rows) — a single wave issues one instruction per ~5 cycles, so every
//     instruction it does not execute is time: ~8.5 x 5 cycles x T rows = 75 us at T = 4096.
// Returns are bit-identical to the lane = column scan (same float32 operations in the same order).
constexpr int kStreamCols = 16, kStreamRows = 64, kStreamDepth = 4;
constexpr int kStreamThreads = 512;      // wave 0: the chain, alone on its SIMD; waves 1, 2, 3, 5: helpers; 4, 6, 7 idle
constexpr int kStreamGroups = kStreamRows / 4;                       // groups of four rows
// One chunk in LDS: six planes [group of four rows][column][row in the group] — (1 - lambda) nv, r,
// 1 - resets, resets nv, 1 - terminations, values.  A ds_read costs a lone wave ~19 cycles whatever
// it fetches (scripts/ubench/chain.hip: 7.3 ns per row from registers, 19.5 ns with one 16-byte
// read per row), so every operand comes as ONE 16-byte read per FOUR rows, and a plain chunk —
// nobody resets or terminates — reads only the first two planes.
// (a row group is 17 sixteen-byte slots apart, not 16: the loaders of gae_stream16_kernel hold
//  4 rows x 4 columns per lane and write one slot per column — with 16 the 64 lanes of a store hit
//  8 of the 32 banks)
constexpr int kStreamGroupPitch = kStreamCols + 1;                  // 16-byte slots per row group
constexpr int kStreamPlane = kStreamGroups * kStreamGroupPitch * 4; // floats
enum StreamPlane : int { P_C1 = 0, P_R = 1, P_M1 = 2, P_C2 = 3, P_M2 = 4, P_V = 5 };
constexpr int kStreamSlot = 6 * kStreamPlane;
constexpr int kStreamLds = kStreamDepth * kStreamSlot + 3 * kStreamPlane +
                          kStreamDepth * kStreamGroups;          // + ret ring, dump, plain-group flags

// Barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier and
// the fence waits for every outstanding GLOBAL access too (s_waitcnt vmcnt(0)): the helper waves
// would then sit out a full HBM round trip per chunk for loads they only need two chunks later.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The chain wave of the streamed scans below (see gae_stream_kernel).  flag_sets: how many words per
// row group the helpers publish (the group is plain when all of them are set).
template <int kFlagSets>
__device__ __forceinline__ void stream_chain(const GaeArgs& g, float* lds, float* ret_ring, int tid,
                                             int64_t w0, int chunks) {
  const int col = tid & (kStreamCols - 1);
  const int64_t wc = min(w0 + col, g.W - 1);
  float last = g.next_values[(g.T - 1) * g.W + wc];               // utils.py:11
  const float lambda = g.lambda, gamma = g.gamma;
  __syncthreads();                                                // chunks 0 and 1 are in LDS
  for (int i = 0; i < chunks; ++i) {
    const float* slot = lds + (i % kStreamDepth) * kStreamSlot;
    const f32x4* planes = reinterpret_cast<const f32x4*>(slot) + col;            // [plane][group][col]
    // (lanes 16 .. 63 mirror the 16 columns; their copies of the returns go to a dump area so that
    //  the stores need no lane mask: saving / restoring exec costs three instructions per group)
    f32x4* out = reinterpret_cast<f32x4*>(tid < kStreamCols ? ret_ring + (i & 1) * kStreamPlane
                                                            : ret_ring + 2 * kStreamPlane) + col;
    // one word per row group, written by the helpers: non-zero = the four rows are plain (below);
    // lane l fetches word l & 15, the ballot makes the mask scalar
    const int* flags = reinterpret_cast<const int*>(ret_ring + 3 * kStreamPlane) +
                       (i % kStreamDepth) * kStreamGroups;
    bool group_plain = flags[col] != 0;
    if (kFlagSets == 2) group_plain = group_plain && flags[kStreamDepth * kStreamGroups + col] != 0;
    const unsigned plain = (unsigned)__ballot(group_plain) & 0xffffu;
    // Straight-line code per chunk — a taken branch costs a single wave more than a row of the
    // chain (instruction fetch restarts), so the choice between the two forms of the recurrence
    // is made ONCE per chunk of 64 rows: `walk<true>` when every row group of the chunk is plain
    // (nobody resets or terminates: 1 - resets == 1, resets * nv == 0, 1 - terminations == 1
    // everywhere, checked on the operands themselves — lines 15-17 of utils.py then return their
    // input, x * 1 and x + 0: exact up to the sign of a zero).  Operands of three row groups live
    // in registers: group g is computed from set g % 3 while the reads of group g + 2 are in
    // flight; they are requested INSIDE the dependent chain (sched_group_barrier pins the
    // interleaving), where an independent instruction issues in the shadow of a stall.
    auto walk = [&](auto plain_chunk) {
      constexpr bool kPlain = decltype(plain_chunk)::value;
      constexpr int kPlanes = kPlain ? 2 : 5;
      f32x4 op[3][5];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int k = 0; k < kPlanes; ++k) op[b][k] = planes[(k * kStreamGroups + b) * kStreamGroupPitch];
      }
#pragma unroll
      for (int grp = 0; grp < kStreamGroups; ++grp) {
        const int cur = grp % 3, nxt = (grp + 2) % 3;
        if (grp + 2 < kStreamGroups) {
#pragma unroll
          for (int k = 0; k < kPlanes; ++k)
            op[nxt][k] = planes[(k * kStreamGroups + grp + 2) * kStreamGroupPitch];
        }
        f32x4 ret;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float boot = op[cur][P_C1][p] + lambda * last;            // utils.py:13-14
          if (!kPlain) {
            boot = boot * op[cur][P_M1][p];                         // :15
            boot = boot + op[cur][P_C2][p];                         // :16
            boot = boot * op[cur][P_M2][p];                         // :17
          }
          last = op[cur][P_R][p] + gamma * boot;                    // :18
          ret[p] = last;
        }
#pragma unroll
        for (int k = 0; k < kPlanes; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x002, kPlain ? 6 : 5, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        out[grp * kStreamGroupPitch] = ret;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (plain == 0xffffu || g.probe == 1) walk(std::true_type{});   // (probe: the chain's own time)
    else walk(std::false_type{});
    lds_barrier();
  }
}

__global__ __launch_bounds__(kStreamThreads) void gae_stream_kernel(GaeArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ double red[4][kStreamThreads / 64];
  double sum = 0.0, sum_sq = 0.0;                                   // (helpers only; the chain: neutral)
  float lo = INFINITY, hi = -INFINITY;
  float* ret_ring = lds + kStreamDepth * kStreamSlot;               // [2][groups][cols][4]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t w0 = (int64_t)blockIdx.x * kStreamCols;
  const int chunks = (int)((g.T + kStreamRows - 1) / kStreamRows);
  // position j of chunk k <-> time row T - 1 - 64 k - j (the chain runs from the end of time)
  if (wave == 0) {
    stream_chain<1>(g, lds, ret_ring, tid, w0, chunks);
  } else if (wave == 4 || wave > 5) {
    // Idle waves: waves land on SIMD (wave % 4), so wave 4 would share the chain's SIMD — every
    // instruction it issued would take an issue slot from the chain.  These waves only keep the
    // barrier count (1 + chunks, like everybody).
    __syncthreads();
    for (int i = 0; i < chunks; ++i) lds_barrier();
  } else {
  // -------------------------------------------------------------------- the helpers
  const int h = (wave == 5 ? 3 : wave - 1) * 64 + (tid & 63);
  const bool probe_idle = g.probe == 1;
  const int col = h & (kStreamCols - 1), sub = h >> 4;              // rows 4 sub .. 4 sub + 3 of a chunk
  const int64_t w = w0 + col, wc = min(w, g.W - 1);
  const bool column = w < g.W;
  // Four register sets used in turn: chunk c travels in set c % 4, requested SIX barriers before the
  // chain needs it (four in flight + two in LDS): ~6 chunk times (8 us) of cover for HBM latency.
  // The loop below is unrolled by four so that the set is a compile-time choice (a run-time index
  // would turn every load into load + select, i.e. a wait).
  struct Rows { float nv[4], rw[4], rs[4], tm[4], vl[4]; } set0, set1, set2, set3;
  auto request = [&](Rows& s, int chunk) {
    if (chunk >= chunks) return;                                    // uniform
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t t = max(g.T - 1 - (int64_t)kStreamRows * chunk - (4 * sub + p), (int64_t)0);
      const int64_t at = t * g.W + wc;
      s.nv[p] = __builtin_nontemporal_load(g.next_values + at);
      s.rw[p] = __builtin_nontemporal_load(g.rewards + at);
      s.rs[p] = __builtin_nontemporal_load(g.resets + at);
      s.tm[p] = __builtin_nontemporal_load(g.terminations + at);
      s.vl[p] = __builtin_nontemporal_load(g.values + at);
    }
  };
  auto publish = [&](const Rows& s, int chunk) {
    if (chunk >= chunks) return;
    float* slot = lds + (chunk % kStreamDepth) * kStreamSlot;
    f32x4 plane[6];
    bool plain = true;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      plane[P_C1][p] = g.one_minus_lambda * s.nv[p];
      plane[P_R][p] = s.rw[p];
      plane[P_M1][p] = 1.f - s.rs[p];
      plane[P_C2][p] = s.rs[p] * s.nv[p];
      plane[P_M2][p] = 1.f - s.tm[p];
      plane[P_V][p] = s.vl[p];
      plain = plain && plane[P_M1][p] == 1.f && plane[P_C2][p] == 0.f && plane[P_M2][p] == 1.f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
      reinterpret_cast<f32x4*>(slot + k * kStreamPlane)[sub * kStreamGroupPitch + col] = plane[k];
    // the 16 columns of this row group are 16 consecutive lanes of one wave
    const unsigned long long votes = __ballot(plain);
    if (col == 0) {
      int* flags = reinterpret_cast<int*>(ret_ring + 3 * kStreamPlane) +
                   (chunk % kStreamDepth) * kStreamGroups;
      flags[sub] = ((votes >> (16 * (sub & 3))) & 0xffffull) == 0xffffull ? 1 : 0;
    }
  };
  auto retire = [&](int chunk) {                                    // returns / advantages of a chunk
    const float* slot = lds + (chunk % kStreamDepth) * kStreamSlot;
    const f32x4 ret = reinterpret_cast<const f32x4*>(
        ret_ring + (chunk & 1) * kStreamPlane)[sub * kStreamGroupPitch + col];
    const f32x4 v4 = reinterpret_cast<const f32x4*>(slot + P_V * kStreamPlane)[sub * kStreamGroupPitch + col];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t t = g.T - 1 - (int64_t)kStreamRows * chunk - (4 * sub + p);
      if (t < 0 || !column) continue;
      const float adv = ret[p] - v4[p];                             // segments.py:42
      __builtin_nontemporal_store(ret[p], g.returns + t * g.W + w);
      __builtin_nontemporal_store(adv, g.advantages + t * g.W + w);
      sum += (double)adv;
      sum_sq += (double)adv * (double)adv;
      lo = fminf(lo, adv);
      hi = fmaxf(hi, adv);
    }
  };
  request(set0, 0);
  request(set1, 1);
  publish(set0, 0);
  publish(set1, 1);
  request(set2, 2);
  request(set3, 3);
  request(set0, 4);
  request(set1, 5);
  __syncthreads();
  auto turn = [&](Rows& s, int i) {             // one barrier interval of the helpers
    if (!probe_idle) {                          // (developer probe: the chain's time alone)
      publish(s, i + 2);                        // requested four barriers ago
      request(s, i + 6);
      if (i > 0) retire(i - 1);
    }
    lds_barrier();
  };
  for (int i = 0; i < chunks; i += 4) {
    turn(set2, i);
    if (i + 1 < chunks) turn(set3, i + 1);
    if (i + 2 < chunks) turn(set0, i + 2);
    if (i + 3 < chunks) turn(set1, i + 3);
  }
  retire(chunks - 1);
  }
  // block partial -> HBM (deterministic: fixed lane / wave order, no atomics)
  sum = wave_sum(sum);
  sum_sq = wave_sum(sum_sq);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if ((tid & 63) == 0) { red[0][wave] = sum; red[1][wave] = sum_sq; red[2][wave] = lo; red[3][wave] = hi; }
  __syncthreads();
  if (tid == 0) {
    double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int i = 1; i < kStreamThreads / 64; ++i) {
      s0 += red[0][i]; s1 += red[1][i];
      mn = fmin(mn, red[2][i]); mx = fmax(mx, red[3][i]);
    }
    const int64_t b = blockIdx.x;
    g.block_sums[4 * b] = s0;
    g.block_sums[4 * b + 1] = s1;
    g.block_sums[4 * b + 2] = mn;
    g.block_sums[4 * b + 3] = mx;
  }
}

// The same pipeline for W % 16 == 0 (every BASELINE size): the helpers move 16 bytes per lane and
// instruction.  With one dword per lane (above) a chunk costs the CU 112 vector-memory instructions
// of 4 x 64 bytes each — at the ~8 ns the memory pipe spends on an instruction whatever it fetches
// that alone is ~1 us per 64 rows, as long as the chain needs.  Here a loader lane holds 4 rows x 4
// columns of every array (20 loads per lane and chunk): ONE wave requests a whole chunk, waves 1
// and 2 take the even and the odd chunks (two register sets each: four chunk times of cover), turn
// them into the chain's operands and publish them, one 16-byte slot [four rows] per column and
// plane; wave 3 retires the chain's returns (two 16-byte stores per row and column quad); wave 0
// is the chain, alone on its SIMD.
typedef float f32x4_row __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kStream16Threads = 256;

__global__ __launch_bounds__(kStream16Threads) void gae_stream16_kernel(GaeArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ret_ring = lds + kStreamDepth * kStreamSlot;               // [2][groups][cols][4] + dump
  int* flag_words = reinterpret_cast<int*>(ret_ring + 3 * kStreamPlane);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int64_t w0 = (int64_t)blockIdx.x * kStreamCols;
  const int chunks = (int)((g.T + kStreamRows - 1) / kStreamRows);
  const int cq = lane & 3, rg = lane >> 2;                          // columns 4 cq .., rows 4 rg ..
  if (wave == 0) {
    stream_chain<1>(g, lds, ret_ring, tid, w0, chunks);
  } else if (wave < 3) {
    // ------------------------------------------------------------------ the two loader waves
    const int mine = wave - 1;                                      // parity of this wave's chunks
    const bool idle = g.probe == 1;                                 // developer probe: barriers only
    struct Rows { f32x4 nv[4], rw[4], rs[4], tm[4], vl[4]; } set0, set1;
    auto request = [&](Rows& s, int chunk) {
      if (chunk >= chunks || idle) return;                          // uniform
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int64_t t = max(g.T - 1 - (int64_t)kStreamRows * chunk - (4 * rg + p), (int64_t)0);
        const int64_t at = t * g.W + w0 + 4 * cq;
        auto load = [&](const float* base) {
          const f32x4_row v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_row*>(base + at));
          return f32x4{v[0], v[1], v[2], v[3]};
        };
        s.nv[p] = load(g.next_values); s.rw[p] = load(g.rewards); s.rs[p] = load(g.resets);
        s.tm[p] = load(g.terminations); s.vl[p] = load(g.values);
      }
    };
    auto publish = [&](const Rows& s, int chunk) {
      if (chunk >= chunks || idle) return;
      float* slot = lds + (chunk % kStreamDepth) * kStreamSlot;
      bool plain = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {                                 // column 4 cq + k: four rows per slot
        f32x4 plane[6];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          plane[P_C1][p] = g.one_minus_lambda * s.nv[p][k];
          plane[P_R][p] = s.rw[p][k];
          plane[P_M1][p] = 1.f - s.rs[p][k];
          plane[P_C2][p] = s.rs[p][k] * s.nv[p][k];
          plane[P_M2][p] = 1.f - s.tm[p][k];
          plane[P_V][p] = s.vl[p][k];
          plain = plain && plane[P_M1][p] == 1.f && plane[P_C2][p] == 0.f && plane[P_M2][p] == 1.f;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
          reinterpret_cast<f32x4*>(slot + q * kStreamPlane)[rg * kStreamGroupPitch + 4 * cq + k] = plane[q];
      }
      // the four column quads of this row group are four consecutive lanes
      const unsigned long long votes = __ballot(plain);
      if (cq == 0)
        flag_words[(chunk % kStreamDepth) * kStreamGroups + rg] =
            ((votes >> (4 * rg)) & 0xfull) == 0xfull ? 1 : 0;
    };
    // chunk c belongs to wave c & 1 and travels in its set (c >> 1) & 1: requested at barrier
    // c - 6, published at c - 2, walked by the chain at c
    request(set0, mine);
    publish(set0, mine);
    request(set1, mine + 2);
    request(set0, mine + 4);
    __syncthreads();
    for (int i = 0; i < chunks; i += 4) {       // barriers i .. i + 3; this wave works at two of them
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (i + k >= chunks) break;
        if ((k & 1) == mine) {
          if (k < 2) { publish(set1, i + k + 2); request(set1, i + k + 6); }
          else { publish(set0, i + k + 2); request(set0, i + k + 6); }
        }
        lds_barrier();
      }
    }
  } else {
    // ------------------------------------------------------------------ the retiring wave
    double sum = 0.0, sum_sq = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    auto retire = [&](int chunk) {                                  // returns / advantages of a chunk
      const float* slot = lds + (chunk % kStreamDepth) * kStreamSlot;
      const f32x4* rets = reinterpret_cast<const f32x4*>(ret_ring + (chunk & 1) * kStreamPlane) +
                          rg * kStreamGroupPitch + 4 * cq;
      const f32x4* vals = reinterpret_cast<const f32x4*>(slot + P_V * kStreamPlane) +
                          rg * kStreamGroupPitch + 4 * cq;
      f32x4 ret[4], val[4];                                         // [column k][row p]
#pragma unroll
      for (int k = 0; k < 4; ++k) { ret[k] = rets[k]; val[k] = vals[k]; }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int64_t t = g.T - 1 - (int64_t)kStreamRows * chunk - (4 * rg + p);
        if (t < 0) continue;
        f32x4_row r, a;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          r[k] = ret[k][p];
          a[k] = ret[k][p] - val[k][p];                             // segments.py:42
          sum += (double)a[k];
          sum_sq += (double)a[k] * (double)a[k];
          lo = fminf(lo, a[k]);
          hi = fmaxf(hi, a[k]);
        }
        const int64_t at = t * g.W + w0 + 4 * cq;
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4_row*>(g.returns + at));
        __builtin_nontemporal_store(a, reinterpret_cast<f32x4_row*>(g.advantages + at));
      }
    };
    __syncthreads();
    for (int i = 0; i < chunks; ++i) {
      if (i > 0 && g.probe == 0) retire(i - 1);
      lds_barrier();
    }
    retire(chunks - 1);
    // block partial -> HBM (deterministic: fixed lane order, no atomics)
    sum = wave_sum(sum);
    sum_sq = wave_sum(sum_sq);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off, 64));
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    if (lane == 0) {
      const int64_t b = blockIdx.x;
      g.block_sums[4 * b] = sum;
      g.block_sums[4 * b + 1] = sum_sq;
      g.block_sums[4 * b + 2] = lo;
      g.block_sums[4 * b + 3] = hi;
    }
  }
}

// {mean, std, all_zero, normalise}: segments.py:43-46 and updaters/actors.py:71.
__device__ __forceinline__ void write_adv_stats(double s0, double s1, double mn, double mx,
                                                double count, float* adv_stats) {
  const double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  // numpy's std of a constant array is exactly 0 (segments.py:44 then skips the
  // normalisation); detect that case by min == max instead of trusting var == 0.
  const bool constant = mn == mx;
  const float std = constant ? 0.f : (float)sqrt(var);
  adv_stats[0] = (float)mean;
  adv_stats[1] = std;
  adv_stats[2] = (constant && mn == 0.0) ? 1.f : 0.f;
  adv_stats[3] = std != 0.f ? 1.f : 0.f;
}

// moments = {sum, sum_sq, -min, max, count}: ranks SUM-reduce [0,1,4] and MAX-reduce [2,3].
__global__ void adv_stats_from_moments_kernel(const double* moments, float* adv_stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    write_adv_stats(moments[0], moments[1], -moments[2], moments[3], moments[4], adv_stats);
}

constexpr int kStatsThreads = 1024;

// Fixed-order fold of the workgroups' partial moments (the segmented pass leaves up to tens of
// thousands of them: 64 threads walking them four dependent loads at a time took longer than the
// scan itself).
__global__ __launch_bounds__(kStatsThreads) void gae_stats_kernel(const double* block_sums,
                                                                  int nblocks, double count,
                                                                  float* adv_stats,
                                                                  double* moments) {
  __shared__ double red[4][kStatsThreads / 64];
  double s0 = 0.0, s1 = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += kStatsThreads) {
    s0 += block_sums[4 * i];
    s1 += block_sums[4 * i + 1];
    mn = fmin(mn, block_sums[4 * i + 2]);
    mx = fmax(mx, block_sums[4 * i + 3]);
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, off, 64));
    mx = fmax(mx, __shfl_xor(mx, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = mn; red[3][wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s0 = 0.0; s1 = 0.0; mn = INFINITY; mx = -INFINITY;
    for (int i = 0; i < kStatsThreads / 64; ++i) {
      s0 += red[0][i]; s1 += red[1][i];
      mn = fmin(mn, red[2][i]); mx = fmax(mx, red[3][i]);
    }
    write_adv_stats(s0, s1, mn, mx, count, adv_stats);
    if (moments != nullptr) {
      moments[0] = s0; moments[1] = s1; moments[2] = -mn; moments[3] = mx; moments[4] = count;
    }
  }
}

std::atomic<int> g_gae_stream{1};   // tuning key "gae_stream": 0 keeps the lane = column scan for one chunk

namespace {

// chunks: 1 = the exact single-chain scan; 0 = the library's choice; > 1 = the segmented pass
// (the segment length is fixed — 128 rows, 64 from W = 8192 — whatever the number asked for).
bool use_onepass(int64_t T, int64_t W, int chunks) {
  if (chunks == 1 || T <= seg_waves(W) * kSegRowsPerWave) return false;
  if (chunks > 1) return true;
  // auto: the single chain needs W alone to fill the chip (64 lanes x 8 rows in flight per wave)
  return W < 65536;
}

// the exact chain for few columns: one workgroup per 16 columns (gae_stream_kernel)
bool use_stream(int64_t T, int64_t W, int chunks) {
  return !use_onepass(T, W, chunks) && W < 16384 && g_gae_stream.load() != 0;
}

struct GaeLayout {
  bool onepass, stream;
  int chunks, tiles;
  int64_t col_blocks, scan_blocks;
  int64_t off_a, off_b, off_incl, off_flags, off_sums, bytes, flag_bytes;
};

GaeLayout gae_layout(int64_t T, int64_t W, int chunks) {
  GaeLayout l{};
  l.onepass = use_onepass(T, W, chunks);
  const int64_t seg_rows = seg_waves(W) * kSegRowsPerWave;
  l.chunks = l.onepass ? (int)((T + seg_rows - 1) / seg_rows) : 1;
  l.tiles = (int)((W + 63) / 64);
  l.col_blocks = (W + kGaeThreads - 1) / kGaeThreads;
  l.stream = use_stream(T, W, chunks);
  l.scan_blocks = l.onepass ? (int64_t)l.chunks * l.tiles
                  : l.stream ? (W + kStreamCols - 1) / kStreamCols : l.col_blocks;
  const int64_t cw = round_up((int64_t)l.chunks * W * 4, 256);
  l.off_a = 0;
  l.off_b = cw;
  l.off_incl = 2 * cw;
  l.off_flags = 3 * cw;
  l.flag_bytes = round_up(((int64_t)l.chunks * l.tiles + 1) * 4, 256);      // flags + the ticket
  l.off_sums = l.off_flags + l.flag_bytes;
  l.bytes = l.off_sums + round_up(l.scan_blocks * 32, 256);
  return l;
}

}  // namespace
}  // namespace tonic

using namespace tonic;

extern "C" int64_t tonic_gae_workspace_bytes(int64_t T, int64_t W, int32_t chunks) {
  if (T <= 0 || W <= 0) return 0;
  return gae_layout(T, W, chunks).bytes;
}

extern "C" int tonic_gae_lambda_returns(const float* d_next_values, const float* d_rewards,
                                        const float* d_resets, const float* d_terminations,
                                        const float* d_values, float* d_returns,
                                        float* d_advantages, float* d_adv_stats,
                                        double* d_adv_moments, int64_t T,
                                        int64_t W, double discount_factor, double trace_decay,
                                        int32_t chunks, void* d_workspace,
                                        int64_t workspace_bytes, void* stream) {
  TONIC_REQUIRE(d_next_values && d_rewards && d_resets && d_terminations && d_values &&
                    d_returns && d_advantages && d_adv_stats,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_gae_lambda_returns: null pointer");
  TONIC_REQUIRE(T > 0 && W > 0 && chunks >= 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_gae_lambda_returns: T=%lld W=%lld chunks=%d", (long long)T,
                (long long)W, chunks);
  const GaeLayout l = gae_layout(T, W, chunks);
  TONIC_REQUIRE(d_workspace && workspace_bytes >= l.bytes, TONIC_ERR_WORKSPACE,
                "gae workspace too small: %lld < %lld", (long long)workspace_bytes,
                (long long)l.bytes);
  char* ws = static_cast<char*>(d_workspace);
  GaeArgs g;
  g.next_values = d_next_values; g.rewards = d_rewards; g.resets = d_resets;
  g.terminations = d_terminations; g.values = d_values; g.returns = d_returns;
  g.advantages = d_advantages;
  g.carry_a = reinterpret_cast<float*>(ws + l.off_a);
  g.carry_b = reinterpret_cast<float*>(ws + l.off_b);
  g.block_sums = reinterpret_cast<double*>(ws + l.off_sums);
  g.T = T; g.W = W;
  // Python-float scalars are rounded to float32 when they multiply a float32 array; `1 -
  // trace_decay` is formed in float64 first (utils.py:14), hence the double arguments.
  g.gamma = (float)discount_factor; g.lambda = (float)trace_decay;
  g.one_minus_lambda = (float)(1.0 - trace_decay);
  g.probe = g_gae_stream.load() == 2 ? 1 : g_gae_stream.load() == 4 ? 2 : 0;   // 4: loaders only (no retiring)
  hipStream_t st = as_stream(stream);
  if (l.onepass) {
    GaeOnePass p{};
    p.g = g;
    p.inclusive = reinterpret_cast<float*>(ws + l.off_incl);
    p.ticket = reinterpret_cast<unsigned*>(ws + l.off_flags);
    p.tiles = l.tiles; p.segments = l.chunks;
    // the maps and the handed-on carries start EMPTY (all-ones words), the ticket at zero
    if (hipMemsetAsync(ws + l.off_a, 0xFF, (size_t)(l.off_flags - l.off_a), st) != hipSuccess ||
        hipMemsetAsync(p.ticket, 0, (size_t)l.flag_bytes, st) != hipSuccess) {
      set_error("tonic_gae_lambda_returns: hipMemsetAsync failed");
      return TONIC_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)(l.chunks * l.tiles));
    if (seg_waves(W) == 4) hipLaunchKernelGGL(gae_onepass_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gae_onepass_kernel<8>, grid, dim3(512), 0, st, p);
  } else if (l.stream) {
    static const bool lds_ok = hipFuncSetAttribute(
        reinterpret_cast<const void*>(gae_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        kStreamLds * (int)sizeof(float)) == hipSuccess;
    TONIC_REQUIRE(lds_ok, TONIC_ERR_LAUNCH, "gae_stream_kernel: %d bytes of LDS refused",
                  kStreamLds * (int)sizeof(float));
    static const bool lds16_ok = hipFuncSetAttribute(
        reinterpret_cast<const void*>(gae_stream16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        kStreamLds * (int)sizeof(float)) == hipSuccess;
    TONIC_REQUIRE(lds16_ok, TONIC_ERR_LAUNCH, "gae_stream16_kernel: %d bytes of LDS refused",
                  kStreamLds * (int)sizeof(float));
    if (W % kStreamCols == 0 && g_gae_stream.load() != 3)
      hipLaunchKernelGGL(gae_stream16_kernel, dim3((unsigned)l.scan_blocks), dim3(kStream16Threads),
                         kStreamLds * sizeof(float), st, g);
    else
      hipLaunchKernelGGL(gae_stream_kernel, dim3((unsigned)l.scan_blocks), dim3(kStreamThreads),
                         kStreamLds * sizeof(float), st, g);
  } else {
    hipLaunchKernelGGL(gae_scan_kernel, dim3((unsigned)l.col_blocks), dim3(kGaeThreads), 0, st, g);
  }
  hipLaunchKernelGGL(gae_stats_kernel, dim3(1), dim3(kStatsThreads), 0, st, g.block_sums,
                     (int)l.scan_blocks, (double)T * (double)W, d_adv_stats, d_adv_moments);
  TONIC_CHECK_LAUNCH("tonic_gae_lambda_returns");
  return TONIC_OK;
}

extern "C" int tonic_advantage_stats_from_moments(const double* d_adv_moments,
                                                  float* d_adv_stats, void* stream) {
  TONIC_REQUIRE(d_adv_moments && d_adv_stats, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_advantage_stats_from_moments: null pointer");
  hipLaunchKernelGGL(adv_stats_from_moments_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                     d_adv_moments, d_adv_stats);
  TONIC_CHECK_LAUNCH("tonic_advantage_stats_from_moments");
  return TONIC_OK;
}
