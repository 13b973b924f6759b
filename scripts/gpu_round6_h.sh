#!/bin/bash
# Round 6, session H: the previous transition's store riding in the acting launch (tonic_q_store_t), q_act through the
# vectorcall shim: off-policy tests, learning curves, the loops, the step probe.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_learning.py -q --timeout 600 -p no:cacheprovider -k "SAC or TD3 or DDPG or D4PG or MPO" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_multirank.py -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
echo "== off-policy loops"
timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h_offpolicy_loops.txt
echo "== step probe"
timeout 600 python scripts/offpolicy_step_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h_offpolicy_step_probe.txt
