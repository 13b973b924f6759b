#!/bin/bash
# Round 5: the push transport (collector transport 3) — its tests, the stress of the completion / freshness
# protocol, and the collect loop with transport 2 and 3 alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_collector.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r05_push_tests.log
tail -5 gpurun_out/r05_push_tests.log
{
for rep in 1 2 3; do
  for t in "2 48" "3 48" "3 0"; do
    set -- $t
    echo "== TONIC_AMD_COLLECTOR_TRANSPORT=$1 TONIC_AMD_COLLECTOR_PUSH_ROWS_KB=$2"
    TONIC_AMD_COLLECTOR_PUSH_ROWS_KB=$2 TONIC_AMD_COLLECTOR_TRANSPORT=$1 timeout 300 python scripts/host_loop_probe.py 3000 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r05_push_host_loop.txt 2>&1
cat gpurun_out/r05_push_host_loop.txt
