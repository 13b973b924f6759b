#!/bin/bash
# Round 5: the per-step entry points through the vectorcall shim (tonic_amd/_fastcall) or through ctypes,
# alternating on one box; then the collector / learning tests on the shim.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for f in 0 1; do
    echo "== TONIC_AMD_FASTCALL=$f"
    TONIC_AMD_FASTCALL=$f timeout 300 python scripts/host_loop_probe.py 3000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260
  done
done
} > gpurun_out/r05_fastcall_host_loop.txt 2>&1
cat gpurun_out/r05_fastcall_host_loop.txt
timeout 900 python -m pytest tests/test_gpu_collector.py tests/test_gpu_learning.py -m gpu -q -x 2>&1 | tail -2
