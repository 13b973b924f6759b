// The per-environment-step collect kernel (mlp64x16.hip) as seen by its two callers: the
// device-resident entry points in mlp64x16.hip and the pinned-host collector (collector.hip).
#pragma once
#include "common.h"

namespace tonic {

// Pre-permuted MFMA operand image of the PPO actor (tonic_ppo_pack_actor).
struct PackedActor {
  int W1I, W2S, B1P, B2P, W3P, HC, total;
  __host__ __device__ PackedActor(int ks1, int ap) {
    W1I = 0; W2S = W1I + 4 * ks1 * 64; B1P = W2S + 4096; B2P = B1P + 64; W3P = B2P + 64;
    HC = W3P + ap * 64; total = HC + 64;
  }
};

constexpr int64_t kRecordFromSegment = 16384;     // floats of one step's observations

struct Collect16Args {
  const float* packed; const float* obs; const float* eps;
  const float* next_obs; const float* rewards; const float* resets; const float* terminations;
  float* seg_obs; float* seg_act; float* seg_next; float* seg_rew; float* seg_rst;
  float* seg_term; float* seg_lp;
  float* norm_acc; float* actions_out;
  int64_t row, W;
  // MeanStd.record running sums: read at norm_acc + row * norm_stride, written at
  // norm_acc + (row + 1) * norm_stride.  0 (device-resident callers): in place.  2 * O (the
  // pinned-host collector): a history with one entry per Segment row, which makes a step
  // idempotent — issuing the same row again recomputes the same sums (see collector.hip).
  int64_t norm_stride;
  int O, A;
  // inputs of the NEXT step (null: none): touched early so that the next launch finds them in
  // L2 / Infinity Cache instead of paying an HBM round trip on its critical path
  const float* pf_eps; const float* pf_next_obs; const float* pf_rewards;
  const float* pf_resets; const float* pf_terminations;
  // Segment row that receives the transition outcome (next observations, rewards, flags).
  // Device-resident collectors know the outcome of step `row` at launch time (outcome_row = row);
  // with the host in the loop the outcome of step t only exists after the environment has
  // consumed the actions, so the launch of step t stores the outcome of step t - 1
  // (outcome_row = row - 1) and none at all for the first step of a rollout (outcome_row < 0).
  int64_t outcome_row;
  // Host-visible completion (null: none): workgroup b stores `done_seq` into done_flags[b]
  // (pinned host memory) once all of its reads of the pinned block and its writes to it are
  // complete; the host spins on those words instead of synchronising the stream.
  unsigned* done_flags; unsigned done_seq;
  // Many workers (the step's observations are more than a staging tile: W * O >= kRecordFromSegment):
  // MeanStd.record does not pull the whole observation block over PCIe a second time — at 1 280
  // workers the step is bound by PCIe bytes (obs + eps for the actor tiles, obs again for the record,
  // next obs for the outcome: 38 us of a 42 us round trip were the record role) — but reads the
  // Segment row the actor tiles have just written: tile_done[b] (device memory) = done_seq once
  // workgroup b's row stores are released.  null: the record reads the pinned block itself.
  unsigned* tile_done;
  // Resident form only (null otherwise): the kernel's park notice.  A slot that waits for the rows of
  // other slots (the record) gives the command up when the notice is out — the workgroups it waits
  // for may have left; the host starts the kernel again and every slot of the command is run anew.
  const unsigned* abandon;
  // ... and the outcome's next observations do not cross PCIe a second time either when the
  // environment has promised (tonic_collector_block_carry_over) that a worker's next observation IS
  // its observation of the following step unless it reset: the actor tiles, which hold this step's
  // observation rows anyway, write them to the PREVIOUS row's next_observations for the workers
  // whose reset flag is 0; the copy role fetches only the rows of the workers that did reset.
  int next_from_obs;
  // Developer probe (null: off; TONIC_AMD_COLLECTOR_STAMPS=1): 100 MHz wall-clock stamps of one
  // actor workgroup, the record workgroup and one copy workgroup, summed per phase:
  // stamps[role * 8 + phase] += now - t0 (t0 = the moment the role saw its command), [.. + 7] counts.
  unsigned long long* stamps; unsigned long long stamp_t0;
};

// What the resident form of the collect kernel needs besides the per-step arguments.
struct CollectResident {
  const unsigned long long* command;   // pinned host memory: the host's next command word
  unsigned* relay;                     // device memory, zero at launch: the park notice among the workgroups
  unsigned* claims;                    // device memory, zero at launch: [slot] = newest command claimed for it
  int poll_sleep;                      // pause between two polls, units of ~0.25 us
  unsigned* parked;                    // pinned host memory: the command the kernel was waiting for when it parked
  const float* eps0; const float* eps1;
  unsigned first_seq;
  unsigned long long park_ticks;       // 100 MHz ticks without a command before parking
};

// One environment step of the pinned-host collector for shapes beyond the fused act kernel
// (O <= 384, A <= 32; mlpwide.hip): ingest (inputs over PCIe once -> device staging + Segment row,
// the previous step's outcome -> its Segment row, MeanStd.record) -> three dense launches -> sample +
// store + actions back to the block + completion words.
struct WideCollect {
  const float* params;                                   // flat PPO actor parameters
  const float* obs; const float* eps;                    // the block's fields (mapped); eps null = greedy
  const float* next_obs; const float* rewards; const float* resets; const float* terminations;
  float* seg_obs; float* seg_act; float* seg_next; float* seg_rew; float* seg_rst; float* seg_term;
  float* seg_lp;
  float* norm_hist;                                      // [rows + 1][2 O] or null
  float* actions_out;                                    // the block's action field (mapped)
  unsigned* done_flags; unsigned done_seq;               // wide_collect_words(W) completion words
  int64_t row, outcome_row, W;
  int O, A;
};
int64_t wide_collect_workspace_bytes(int64_t W, int O, int A);
int wide_collect_words(int64_t W);
int wide_collect_step(const WideCollect& c, void* d_workspace, int64_t workspace_bytes,
                      hipStream_t stream);

int collect16_ks1(int O);
int collect16_ap(int A);
int collect16_blocks(int64_t W);          // workgroups (= completion words) of one launch
int launch_collect16(const Collect16Args& c, hipStream_t stream);
int launch_collect_resident(const Collect16Args& c, const CollectResident& r, hipStream_t stream);
int launch_actor_pack(const float* d_actor_params, float* d_packed, int O, int A,
                      hipStream_t stream);

}  // namespace tonic
