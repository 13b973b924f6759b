#!/bin/bash
# Round 6, session D: off-policy acting + store on the environment's block (tonic_collector_q_act): the drop-in
# trajectory tests, the loops on their own metric with the block path on / off, the column-slice micro-benchmark
# with all of a thread's polls in flight together, the whole GPU suite.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
echo "== drop-in trajectories"
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider -x -k "drop_in or update_matches" 2>&1 | tail -30
echo "== column slices"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $REPO/scripts/ubench/col_slice.hip -o /tmp/col_slice && timeout 120 /tmp/col_slice | tee $REPO/gpurun_out/r06d_col_slice.txt
echo "== off-policy loops, block path"
timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_offpolicy_loops_block.txt
echo "== off-policy loops, staged copies (TONIC_AMD_Q_BLOCK=0)"
TONIC_AMD_Q_BLOCK=0 timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_offpolicy_loops_staged.txt
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rf 2>&1 | tail -30 | tee gpurun_out/r06d_pytest_gpu.log
