#!/bin/bash
# Round 5: after the last collector change — its tests again, and the functional two-rank PPO line (two processes
# on ONE device, gloo between them) with the push transport.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_collector.py tests/test_gpu_multirank.py tests/test_gpu_learning.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r05d_tests.log
cat gpurun_out/r05d_tests.log
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r05_bench_gpus2_shared_device.json 2> gpurun_out/r05d_bench.err
tail -c 900 gpurun_out/r05_bench_gpus2_shared_device.json; echo; tail -3 gpurun_out/r05d_bench.err
