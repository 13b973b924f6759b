#!/bin/bash
# Round 4, GPU session B: the collector's new protocol first (a hang here must not eat the session),
# then the multi-rank overlap, then everything.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== collector: absent workgroups"
timeout 300 python -m pytest -q --timeout 120 -p no:cacheprovider -x \
  "tests/test_gpu_collector.py::test_resident_kernel_runs_the_slots_of_absent_workgroups" \
  "tests/test_gpu_collector.py::test_resident_kernel_parks_and_resumes" \
  "tests/test_gpu_collector.py::test_collector_steps_match_the_oracle" \
  "tests/test_gpu_collector.py::test_completion_words_order_the_actions" \
  2>&1 | tail -40 | tee gpurun_out/r04b_collector.log
echo "== overlap"
timeout 400 python -m pytest -q --timeout 200 -p no:cacheprovider \
  "tests/test_gpu_collector.py::test_critic_iterations_under_the_next_rollout_are_bit_identical" \
  "tests/test_gpu_collector.py::test_critic_chain_under_a_running_rollout_is_bit_identical_at_size" \
  "tests/test_gpu_multirank.py::test_two_ranks_keep_the_critic_under_the_next_rollout" \
  -s 2>&1 | tail -40 | tee gpurun_out/r04b_overlap.log
