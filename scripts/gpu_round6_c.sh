#!/bin/bash
# Round 6, session C: kernel statistics of the SAC update with the weight images, the column-slice micro-benchmark,
# the off-policy loops on their own metric, the multi-rank tests (the phases rebuild the images every call).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
echo "== column slices (scripts/ubench/col_slice.hip)"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $REPO/scripts/ubench/col_slice.hip -o /tmp/col_slice && timeout 120 /tmp/col_slice | tee $REPO/gpurun_out/r06c_col_slice.txt
echo "== SAC update, weight images: kernel statistics"
bash $REPO/scripts/gpu_profile_sac.sh graph
cp $REPO/gpurun_out/sac_kernel_stats.csv $REPO/gpurun_out/r06c_sac_kernel_stats_images.csv
echo "== SAC update, float32 passes: kernel statistics"
TONIC_AMD_TUNING=q_images=0 bash $REPO/scripts/gpu_profile_sac.sh graph
cp $REPO/gpurun_out/sac_kernel_stats.csv $REPO/gpurun_out/r06c_sac_kernel_stats_f32.csv
cd $REPO
echo "== off-policy loops"
timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c_offpolicy_loops.txt
echo "== multi-rank + off-policy tests"
timeout 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8
