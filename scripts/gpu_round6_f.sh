#!/bin/bash
# Round 6, session F: the update call in chunks (host draws under the GPU's work): bit-identity, the loops.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider -k "chunks or drop_in or full_size" 2>&1 | tail -15
echo "== off-policy loops, chunked update calls"
timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f_offpolicy_loops_chunked.txt
echo "== off-policy loops, one graph per update call (TONIC_AMD_UPDATE_CHUNK=0)"
TONIC_AMD_UPDATE_CHUNK=0 timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f_offpolicy_loops_one_graph.txt
