#!/bin/bash
# Round 6, session A (gpurun -- bash scripts/gpu_round6_a.sh): the whole -m gpu suite (all failures, not -x), smoke, the
# default bench line — the state after the GAE default / collector probe / ADVICE changes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rf 2>&1 | tail -60 | tee gpurun_out/r06a_pytest_gpu.log
echo "== the out-of-envelope test, verbose"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -s -k "non_finite" -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r06a_nonfinite.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"
timeout 900 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_bench.json; tail -5 gpurun_out/r06a_bench.err
