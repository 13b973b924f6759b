#!/bin/bash
# One GPU-box session (gpurun -- bash scripts/gpu_session.sh): the foreign-kernel tests three times, the whole -m gpu suite, smoke, the default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python -m pytest -q --timeout 120 -p no:cacheprovider "tests/test_gpu_collector.py::test_resident_kernel_runs_the_slots_of_absent_workgroups" 2>&1 | tail -3; done
echo "== pytest -m gpu (all)"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/session_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"
timeout 900 python bench.py > gpurun_out/session_bench.json 2> gpurun_out/session_bench.err
tail -c 400 gpurun_out/session_bench.json; tail -5 gpurun_out/session_bench.err
