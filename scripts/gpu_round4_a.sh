#!/bin/bash
# Round 4, GPU session A: the whole -m gpu suite (new at-size tests first), smoke, the default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new tests"
timeout 600 python -m pytest -q --timeout 500 -p no:cacheprovider \
  "tests/test_gpu_collector.py::test_critic_chain_under_a_running_rollout_is_bit_identical_at_size" \
  "tests/test_gpu_collector.py::test_critic_iterations_under_the_next_rollout_are_bit_identical" \
  "tests/test_gpu_parity.py::test_two_whole_iterations_at_baseline_size_vs_oracle" \
  "tests/test_gpu_parity.py::test_value_regression_grad_width_is_an_argument" \
  "tests/test_gpu_offpolicy.py::test_buffer_gather_bit_exact_at_baseline_size" \
  "tests/test_gpu_offpolicy.py::test_a_lost_workgroup_of_a_chained_launch_skips_the_step_and_raises" \
  -s 2>&1 | tail -60 | tee gpurun_out/r04a_new_tests.log
echo "== pytest -m gpu (all)"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=15 2>&1 | tail -70 | tee gpurun_out/r04a_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/r04a_smoke.log
echo "== bench"
timeout 900 python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 600 gpurun_out/r04a_bench.json; tail -5 gpurun_out/r04a_bench.err
