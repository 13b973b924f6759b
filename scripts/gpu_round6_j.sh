#!/bin/bash
# Round 6, session J: TD3's policy passes of iteration k + 1 as more workgroups of the critic-step launch of a not-due
# iteration k (tonic_q_iteration_t.ahead / .stage / .slot, DDPG._enqueue_fused_iterations): tests, rates.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -15
echo "== rates (ahead)"
timeout 600 python scripts/offpolicy_rates.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee gpurun_out/r06j_offpolicy_rates.txt
echo "== rates (TONIC_AMD_POLICY_AHEAD=0)"
TONIC_AMD_POLICY_AHEAD=0 timeout 600 python scripts/offpolicy_rates.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee gpurun_out/r06j_offpolicy_rates_serial.txt
