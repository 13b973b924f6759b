#!/bin/bash
# Round 5, last call: the GPU tier and smoke() on the final build.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_gpu_tests.log
tail -3 gpurun_out/r05_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r05_smoke.log
