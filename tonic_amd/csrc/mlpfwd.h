// Fused forward of a two-hidden-layer ReLU MLP with up to two narrow heads: see mlpfwd.hip.
#pragma once
#include <atomic>
#include "gemm16.h"
#include "bufstore.h"

namespace tonic {

// Parameter layout of the off-policy networks (see include/tonic_hip.h, "off-policy parameter
// layout"): tensors in parameters() order, every tensor starts on a 16-byte boundary, and the
// rows of a [rows, cols] weight are weight_ld(cols) floats apart — a multiple of 4 floats that
// is NOT a multiple of 32.  The kernels stream MFMA operands straight from L2, lane = row: with
// rows a multiple of 128 bytes apart all 16 rows of a load hit the same L1 bank group and the
// stream runs at 38 B/ns per CU instead of 70 - 84 (profiles/r01_ubench_row_stride.md), and
// rows that are not 16-byte aligned split every 16-byte load.  The padding floats are zero and
// stay zero (zero gradient -> Adam moves nothing; polyak of zeros).
__host__ __device__ inline int weight_ld(int cols) {
  const int ld = (cols + 3) / 4 * 4;
  return ld % 32 == 0 ? ld + 4 : ld;
}
__host__ __device__ inline int64_t slot4(int64_t floats) { return (floats + 3) / 4 * 4; }

// ---- what follows a policy's heads (shared by the stand-alone kernels of offpolicy.hip and the
//      tail of mlp_forward_kernel, so that both give the same bits)
constexpr float kSacLogEps = 1e-6f;            // actors.py:15
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// One action of SquashedMultivariateNormalDiag.rsample_with_log_prob (actors.py:11-16,94-98):
// sigma = clamp(softplus(spre), 1e-4, 1), u = loc + eps * sigma, a = tanh(u),
// term = N(u; loc, sigma).log_prob - log(1 - a^2 + 1e-6).
struct SquashedSample { float action, sigma, logp_term; };
__device__ __forceinline__ SquashedSample squashed_sample(float loc, float spre, float eps,
                                                          bool has_eps) {
  SquashedSample r;
  const float raw = softplus_f(spre);
  r.sigma = fminf(fmaxf(raw, 1e-4f), 1.0f);
  const float u = has_eps ? loc + eps * r.sigma : loc;             // rsample: loc + eps * scale
  r.action = tanhf(u);
  const float d = u - loc;
  const float normal = -(d * d) / (2.f * (r.sigma * r.sigma)) - logf(r.sigma) - kHalfLog2Pi;
  r.logp_term = normal - logf(1.f - r.action * r.action + kSacLogEps);
  return r;
}

// TD3 target action: clamp(a + clamp(scale * eps, -clip, clip), -1, 1)  (critics.py:130-134)
__device__ __forceinline__ float noisy_target_action(float action, float eps, float scale,
                                                     float clip) {
  float noise = scale * eps;
  noise = fminf(fmaxf(noise, -clip), clip);
  return fminf(fmaxf(action + noise, -1.f), 1.f);
}

// lanes that share one sample in sac_sample_kernel: the log-probability terms are folded by a
// xor tree over them, which the fused tail re-plays in the same order
__host__ __device__ inline int sample_group(int A) { int g = 1; while (g < A && g < 32) g *= 2; return g; }

enum PolicyPost : int { POST_NONE = 0, POST_SQUASHED_SAMPLE = 1, POST_TARGET_NOISE = 2, POST_COPY = 3 };

// Where value z (a network of the launch) of batch row r lives: z * net + (r / 16) * tile + r % 16.
// Dense arrays [nets][Bp]: {Bp, 16}.  The EXCHANGE LINES of the chained launches (q_chain kernels):
// 128-byte lines per 16-row tile, ONE WRITER per line and ONE WRITE per line between two resets (the
// critic step and the actor step of an iteration have lines of their own: a reader must find a line
// either empty or final) —
//   line z      the 16 values of network z of the launch's first parameter set (the targets of the
//               critic step, the critics of the actor step): {32, 128}
//   line 2 + z  the 16 values of network z of the second set (the online critics): {32, 128} from + 64
// A line is only ever read after its writer has published it, and nobody else writes into it: the
// loads are agent-scope, but a line that sits in an XCD's L2 is served from there — so a line that
// two workgroups share (two tiles' values, or the twin critics' halves) reaches a reader's L2, or the
// L2 of a writer that reads it back, with the other half still unwritten (seen as stale q values
// in the logged sums and in the twin critics' objective when the lines were shared).
struct ValueLines {
  int net, tile;
  __host__ __device__ int index(int z, int r) const { return z * net + (r >> 4) * tile + (r & 15); }
};
constexpr int kExchangeTileFloats = 256;               // 8 lines of 32 floats: 0 - 3 the critic step's, 4 - 5 the actor step's

struct MlpFwdArgs {
  const float* X;            // [B, ldx] inputs, K1 columns used
  int ldx, K1;
  const float* W1; const float* b1;   // [H, K1] rows ldw1 apart, [H]
  const float* W2; const float* b2;   // [H, H] rows ldw2 apart, [H]
  const float* Wh[2]; const float* bh[2];   // heads: [NH, H] rows ldw2 apart, [NH]
  int ldw1, ldw2;            // weight row strides (weight_ld): multiples of 4, 16-byte aligned rows
  int heads, NH;
  float* h1; float* h2;      // [B, H] hidden activations, rows ldh apart (written: the backward needs them)
  int ldh;                   // >= H, multiple of 4 (weight_ld(H): the weight-gradient GEMMs walk these rows)
  float* out[2];             // head outputs [B, ldo]
  int ldo;
  int act[2];                // GemmAct per head (ACT_NONE | ACT_TANH)
  int B, H;
  // batch of networks over blockIdx.y: element strides between consecutive networks
  int64_t stride_params, stride_hidden, stride_out;
  // networks >= split read a second parameter set and a second input (target critics on
  // (s', a') and online critics on (s, a) in one launch): parameter pointers move by
  // second_params ELEMENTS on top of the linear stride, the input is X2.  split >= nets: unused.
  int split;
  int64_t second_params;
  const float* X2;
  // post != POST_NONE (single network, NH <= 64): what follows the heads runs in the same launch.
  //   POST_SQUASHED_SAMPLE (heads = 2): actions / sigma [B, NH] dense, log-probabilities [B]
  //   POST_TARGET_NOISE    (heads = 1, tanh head): actions = noisy_target_action(head, eps)
  //   POST_COPY            (heads = 1): dense actions out of the padded head buffer
  int post;
  const float* post_eps;     // [B, NH] standard-normal draws (SAMPLE: may be null = greedy)
  float* post_actions;       // [B, NH]
  float* post_sigma;         // [B, NH] or null
  float* post_logp;          // [B] or null
  float noise_scale, noise_clip;
  // enc_out != null: the tail also writes the critics' input rows (ObservationActionEncoder,
  // encoders.py:28-31)  enc_out[row] = [ (enc_obs[row] - mean) / std , post_actions[row] ],
  // rows enc_ld apart, and — enc_out2 != null — a second pair built from stored actions,
  // enc_out2[row] = [ (enc_obs2[row] - mean) / std , enc_act2[row] ].
  const float* enc_obs; const float* enc_obs2; const float* enc_act2;   // [B, enc_O], [B, NH]
  const float* enc_mean; const float* enc_std;                          // [enc_O]
  float enc_clip;                                                       // MeanStd(clip): +inf = none
  float* enc_out; float* enc_out2;
  int enc_O, enc_ld;
  float* reset_area; int64_t reset_floats;   // the launch AHEAD of the chained ones: fill with kExchangeEmpty
  unsigned* reset_failed;                    //   ... and clear their failure word
  int reset_blocks;                          //   workgroups that share the fill, numbered net * tiles + tile (0: the
                                             //   whole grid of a launch of its own; set where the pass rides in
                                             //   another launch: QCriticStep::ahead)
  float* xq;                    // chained launches: the value head's outputs ALSO go (agent-scope stores) to the launch's
                                //   exchange lines, see ValueLines; null: none
  int tail_offset;              // set by launch_mlp_forward: where the tail's LDS images start (floats)
  unsigned long long* stamps;   // developer probe (tonic_debug_forward_stamps): null in the product path
  FwdImages img;                // img.block != null: the products run on fp16x2 terms from weight images (mlpimg.h)
  int hidden_from;              // image pass: h1 / h2 go to HBM only for networks >= hidden_from (a network without a
                                //   backward — the targets, the critic step's policy — has no reader for them: at
                                //   B = 1 024 a quarter of the fused iteration's HBM writes)
  // acting on a collector's block (tonic_collector_q_act): every workgroup copies its 16 input rows (K1 columns)
  // to rows_out (rows_ld apart; null: no copy), releases its stores to the SYSTEM (the tail wrote the actions into
  // mapped host memory) and then writes done_seq into done_flags[blockIdx.x] at system scope — the host polls
  // these words (tonic_collector_wait_actions).  Null done_flags: none of this.
  unsigned* done_flags; unsigned done_seq;
  float* rows_out; int rows_ld;
  // ... and (store_on) ONE more workgroup behind the row tiles stores the PREVIOUS step's transition (bufstore.h:
  // Buffer.store + MeanStd.record; its sources are the block's outcome fields, read in place, and the device copy
  // of that step's observation rows) and writes the completion word behind the tiles' — the environment may
  // overwrite the block once every word is out.  lds_floats: the launch's dynamic LDS, the record's staging tile.
  BufferStoreArgs store; int store_on; int lds_floats;
  // tail2.post != POST_NONE (two networks, split == 1): network 1 — the second parameter set on
  // the second input — has a tail of its own with its own outputs (the fused learner iteration
  // runs the policy passes of the critic step AND of the actor step as one launch: SAC the online
  // actor on s' and on s, TD3 the target actor on s' and the online actor on s).  The encoder's
  // statistics (enc_mean / enc_std / enc_clip / enc_O / enc_ld) are shared.
  struct Tail2 {
    int post;
    const float* eps; float* actions; float* sigma; float* logp;
    const float* enc_obs; float* enc_out;
  } tail2;
};

// Input-gradient chain of the same network (see mlp_backward_kernel in mlpfwd.hip).
struct MlpBwdArgs {
  int heads;                 // 0: critic (dq, w3); 1..2: actor heads (dhead, Wh)
  const float* dq;           // [B] per network            (heads == 0)
  const float* w3;           // [H]                        (heads == 0)
  const float* dhead[2];     // [B, ldh] gradients at the head outputs
  const float* Wh[2];        // [NH, H] rows ldw2 apart
  int NH, ldh;
  const float* W2;           // [H, H] rows ldw2 apart
  const float* W1;           // [H, K1] rows ldw1 apart (only for dxa)
  int ldw1, ldw2;
  int K1, xa_first, xa_count;   // dxa = columns [xa_first, xa_first + xa_count) of dz1 . W1
  const float* h1; const float* h2;   // [B, H] forward activations (ReLU masks), rows ldhid apart
  float* dz2; float* dz1;    // [B, H] outputs, rows ldhid apart
  int ldhid;
  float* dxa;                // [B, ldxa]
  int ldxa;
  int B, H;
  int64_t stride_params, stride_hidden, stride_dq, stride_dxa;
  // heads == 0, loss != LOSS_GIVEN: dq is formed HERE from the forward outputs — the element-wise
  // loss of the step folded into its backward launch (one launch and one pass over q less) — and
  // written to dq for the weight-gradient GEMM; workgroup (0, 0) also folds the logged sums.
  //   LOSS_TD     y = r + disc * (min over the target critics - alpha * logp'), dq_z = 2 (q_z - y)
  //               (critics.py:72-79, 166-175, 219-227); stats {sq_err_sum, q1_sum, q2_sum, 0, 0, B, 0, 0}
  //   LOSS_ACTOR  SAC: alpha * logp - min(q1, q2); TD3 / DDPG: -q1 (actors.py:177-179, 254-257);
  //               dq_z = -1 on the smaller critic (-1/2 each on ties); stats {loss_sum, 0, ..., B, ..}
  int loss;
  const float* l_rewards; const float* l_discounts; const float* l_tq; const float* l_logp;
  const float* l_q; float* l_stats;
  float l_alpha;
  int l_nets, l_Bp;
  ValueLines l_tq_at, l_q_at;   // where value z of row r lives in l_tq / l_q (dense: {Bp, 16})
  // heads >= 1, hb_dxa0 != null: the gradients at the head outputs are not given but FORMED here
  // from the critics' action-column input gradients (actor_head_backward_kernel folded into this
  // launch, same expressions -> same bits) and written to dhead[0] / dhead[1] for the
  // weight-gradient GEMM.  hb_sac: squashed-Gaussian head (SAC), else the tanh head (TD3 / DDPG).
  const float* hb_dxa0; const float* hb_dxa1;      // [B, hb_ldxa]; dxa1 null: one critic
  const float* hb_act; const float* hb_eps; const float* hb_sigma;   // [B, NH] dense
  const float* hb_spre;                            // [B, ldh] pre-softplus scale head (SAC)
  int hb_ldxa, hb_sac;
  float hb_alpha;
  unsigned* exchange_failed;    // chained launches: l_tq / l_q / hb_dxa* are exchange words (exchange_read), dxa is
                                //   written with exchange_write; the word a reader sets when a value never came
  BwdImages img;                // img.block != null: the products run on fp16x2 terms from weight images (mlpimg.h)
  int skip_dz;                  // image pass: dz2 / dz1 stay out of HBM (a frozen network's chain: no weight gradients)
};
enum MlpBwdLoss : int { LOSS_GIVEN = 0, LOSS_TD = 1, LOSS_ACTOR = 2 };

// The TD target and the errors of one sample (shared by critic_loss_kernel and the folded form)
// ---- hand-over between the workgroups of a chained launch: "the data is its own flag"
// An exchanged float is written ONCE per launch with an agent-scope store and read with agent-scope
// loads until it is no longer kExchangeEmpty — the pattern every exchanged word holds when the launch
// starts (the launch ahead of the chained ones fills the exchange area with it).  Each location is
// coherent by itself (a relaxed atomic at agent scope); nothing is assumed about the ORDER in which
// two locations become visible.  (An arrival counter behind `s_waitcnt vmcnt(0)` is not enough on
// gfx950: the data's write-through may still be on its way when the counter — an atomic performed at
// the memory side — is already seen: 1 run in 3 of the bit-identity test read a stale q; and a release
// fence that waits for it is an L2 write-back of everything the XCD has dirtied: +13 % per iteration.)
constexpr unsigned kExchangeEmpty = 0x7fa5c3e1u;       // a signalling-NaN pattern no arithmetic produces
constexpr unsigned long long kChainTimeoutTicks = 25000000ull;     // 250 ms of the 100 MHz wall clock

__device__ __forceinline__ float exchange_read(const float* p, unsigned* failed) {
  const unsigned* word = reinterpret_cast<const unsigned*>(p);
  unsigned bits = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bits == kExchangeEmpty) {
    const unsigned long long t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(1);
      bits = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wall_clock64() - t0 > kChainTimeoutTicks) {         // a lost workgroup must not hang the device
        __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    } while (bits == kExchangeEmpty);
  }
  return __uint_as_float(bits);
}
__device__ __forceinline__ void exchange_write(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (`exchange`: tq / q are written by other workgroups of the same launch; null: plain loads)
__device__ __forceinline__ float shared_value(const float* p, unsigned* exchange) {
  return exchange != nullptr ? exchange_read(p, exchange) : *p;
}
// N exchanged values at once: all N loads are issued TOGETHER (one L2 round trip when the writers are done — the
// usual case for everything but the first), and only a word that is still empty is polled.  exchange_read one
// after the other is a round trip EACH (its spin loop is control flow on the loaded value: nothing overlaps):
// eight of them stood at the head of the actor role's chain (round 6).  Same values, same arithmetic.
template <int N>
__device__ __forceinline__ void shared_values(const float* const (&p)[N], float (&v)[N], unsigned* exchange) {
  if (exchange == nullptr) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = *p[i];
    return;
  }
  unsigned bits[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
    bits[i] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int i = 0; i < N; ++i)
    v[i] = bits[i] == kExchangeEmpty ? exchange_read(p[i], exchange) : __uint_as_float(bits[i]);
}
__device__ __forceinline__ float td_target(const float* rewards, const float* discounts,
                                           const float* tq, const float* logp_next, float alpha,
                                           int m, ValueLines at, int nets,
                                           unsigned* coherent = nullptr) {
  if (nets == 1) return rewards[m] + discounts[m] * shared_value(tq + at.index(0, m), coherent);
  const float* const where[2] = {tq + at.index(0, m), tq + at.index(1, m)};
  float both[2];
  shared_values(where, both, coherent);
  float next = fminf(both[0], both[1]);
  if (logp_next) next = next - alpha * logp_next[m];
  return rewards[m] + discounts[m] * next;
}

// d (actor objective) / d q_z of one sample (shared by actor_loss_kernel and the folded form)
__device__ __forceinline__ float actor_dq(const float* q, int m, ValueLines at, int twin, int z,
                                          unsigned* coherent = nullptr) {
  if (!twin) return -1.f;
  const float* const where[2] = {q + at.index(0, m), q + at.index(1, m)};
  float both[2];
  shared_values(where, both, coherent);
  const float q1 = both[0], q2 = both[1];
  if (z == 0) return q1 < q2 ? -1.f : (q1 == q2 ? -0.5f : 0.f);
  return q2 < q1 ? -1.f : (q1 == q2 ? -0.5f : 0.f);
}

// Sum of up to three per-thread values over a workgroup of whole waves: float64 xor tree inside
// each wave, then the wave partials in wave order by thread 0 (deterministic).  Result valid on
// thread 0 only.
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c) {
  __shared__ double wave_part[3][16];
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { wave_part[0][wave] = a; wave_part[1][wave] = b; wave_part[2][wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      a += wave_part[0][w]; b += wave_part[1][w]; c += wave_part[2][w];
    }
  }
}

// ---- several dependent passes over the same 16-row tiles as ONE launch (q_chain kernels, mlpfwd.hip)
// The workgroups of a launch are numbered tile-major — the `roles` workgroups of a tile are
// neighbours in dispatch order and a role only ever waits for values of roles of the same tile (a
// lower index, or — the twin critics of the actor step — its immediate neighbour), so whatever part
// of the grid is resident can always make progress.  What crosses workgroups inside a launch (q
// values, action-column gradients: a few floats per row) goes through the launch's exchange area:
// see exchange_read / ValueLines.  A value that does not come within kChainTimeoutTicks (a lost
// workgroup) does not hang the device: the reader goes on, the iteration's failure word is set, its
// logged losses become NaN and its optimizer epilogues write NOTHING (AdamFold::skip) — parameters,
// moments and targets stay what they were; agents.DDPG._update raises on the row's give-up mark.
// Critic step: roles [target_0 .. target_{nets-1} | online_0 .. online_{nets-1}] (+ one workgroup
// for the logged sums) — the targets' forward on (s', a'), the online critics' forward on (s, a),
// and — with the targets' values of the tile — the TD loss and the online critics' input-gradient chain.
struct QCriticStep {
  MlpFwdArgs fwd;            // 2 nets networks: targets on X, online (second set) on X2; split = nets
  MlpBwdArgs bwd;            // the online critics' chain, loss = LOSS_TD
  int nets;
  int lose_first_target;     // test hook (tuning key "chain_fault"): workgroup 0 returns without a word
  // ahead_nets > 0 (image passes): `ahead_nets` x tiles MORE workgroups behind the step's own run the policy passes
  // (launch 1) of the NEXT iteration — with delayed actor updates (td3.py:43-46) a critic step that is not followed
  // by an actor step leaves everything those passes read alone; they exchange nothing with the step's roles and
  // write another set of buffers (tonic_q_iteration_t.slot), and at B = 100 the step fills a ninth of the chip.
  MlpFwdArgs ahead;
  int ahead_nets;
};
// Actor step: roles [critic_0 .. critic_{used-1} | actor] — the critics' forward on (s, a_new), the
// actor objective (the twin critics exchange q), their chain down to the action columns, then the
// head backward and the actor's chain.
struct QActorStep {
  MlpFwdArgs fwd;            // `used` critics on X3
  MlpBwdArgs bwd;            // their chain, loss = LOSS_ACTOR, dxa
  MlpBwdArgs actor;          // head backward (formed from dxa) + the actor's chain
  int used;
};
int launch_q_critic_step(const QCriticStep& c, hipStream_t stream);
int launch_q_actor_step(const QActorStep& c, hipStream_t stream);
extern std::atomic<int> g_chain_fault;
extern std::atomic<int> g_q_chain;          // tuning key "q_chain": 0 keeps one launch per pass
extern std::atomic<int> g_q_images;         // tuning key "q_images": 0 keeps the float32 passes (no weight images)
bool mlp_image_pass_supported(int K1, int H);

bool mlp_forward_supported(int H, int NH, int heads);
bool mlp_policy_tail_supported(int H, int NH);
extern std::atomic<int> g_policy_tail;      // tuning key "policy_tail": 0 keeps sampling / noise / copy in their own launches
bool mlp_backward_supported(int H, int NH, int heads, int xa_count);
int launch_mlp_backward(const MlpBwdArgs& a, int nets, hipStream_t stream);
int launch_mlp_forward(const MlpFwdArgs& a, int nets, hipStream_t stream);

}  // namespace tonic
