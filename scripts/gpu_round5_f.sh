#!/bin/bash
# Round 5: the resident kernel's stress tests (foreign kernels holding the chip, random bursts) with the inputs pushed.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_collector.py -m gpu -q -x -k "absent_workgroups or random_foreign or parks" 2>&1 | tail -4 | tee gpurun_out/r05f_tests.log
