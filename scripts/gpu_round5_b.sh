#!/bin/bash
# Round 5, second GPU session: the GPU test tier again (the first session stopped at a too-strict bound of a new
# test) and the timing-only knobs of the grad kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05b_tests.log
tail -8 gpurun_out/r05b_tests.log
timeout 300 python scripts/grad_knobs_timing.py 2>&1 | tee gpurun_out/r05b_grad_knobs.txt | tail -14
