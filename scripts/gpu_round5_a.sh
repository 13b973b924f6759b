#!/bin/bash
# Round 5, first GPU session: the whole GPU test tier on the product + dev libraries, the default bench line,
# and the rocprofv3 kernel statistics of the bench command.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05a_tests.log
tail -5 gpurun_out/r05a_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
tail -c 1500 gpurun_out/r05a_bench.json; tail -3 gpurun_out/r05a_bench.err
bash scripts/gpu_profile.sh 2>&1 | tail -16
cp gpurun_out/prof_kernel_stats.csv gpurun_out/r05a_bench_kernel_stats.csv
