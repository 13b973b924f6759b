#!/bin/bash
# Round 5: the push transport (collector transport 3) — the collect loop with the pause between two polls of the
# command word varied (the polls read the GPU's own memory now: nobody's store waits for them).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for s in 1 0 2 4; do
    echo "== TONIC_AMD_COLLECTOR_POLL_SLEEP=$s"
    TONIC_AMD_COLLECTOR_POLL_SLEEP=$s timeout 300 python scripts/host_loop_probe.py 3000 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r05_push_poll_sleep.txt 2>&1
cat gpurun_out/r05_push_poll_sleep.txt
