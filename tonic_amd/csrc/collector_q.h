// What an acting launch on a collector's block needs from the handle (collector.hip) — used by
// tonic_collector_q_act (offpolicy.hip), the off-policy agents' step on the shared block.
#pragma once
#include "common.h"

namespace tonic {

struct CollectorStep {
  int64_t W;
  int O, A;
  const float* observations;       // device views of the block's fields (the page-locked block, mapped)
  const float* eps;                // the noise rows of the slot asked for (null: none)
  float* actions_out;              // where the policy's actions go for the host: the block's second noise field
  // the block's record of the step BEFORE (what the environment wrote around its last step): the executed
  // actions, the next observations, rewards and flags — the sources of the transition's store
  const float* actions; const float* next_observations; const float* rewards; const float* resets;
  const float* terminations;
  unsigned* done_flags;            // completion words (one per 16-row workgroup) ...
  unsigned seq;                    // ... and the value this step writes into them
};

void collector_shape(tonic_collector_t* c, int64_t* W, int* O, int* A);

// Opens a step on collector `c` (nothing else in flight): bumps the sequence number, tells
// tonic_collector_wait_actions how many completion words to expect (one per 16 rows + extra_words).  eps_slot: -1
// none, 0 the first noise field.
int collector_begin_q_step(tonic_collector_t* c, int eps_slot, int extra_words, CollectorStep* out);

}  // namespace tonic
