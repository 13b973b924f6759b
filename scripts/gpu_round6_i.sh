#!/bin/bash
# Round 6, session I: the argument segments of the off-policy learner's kernels requested at their entry
# (kernarg_prefetch, common.h): off-policy tests, rates, the forward stamps.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6
echo "== rates"
timeout 600 python scripts/offpolicy_rates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06i_offpolicy_rates.txt
echo "== stamps"
timeout 600 python scripts/forward_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06i_forward_stamps.txt
