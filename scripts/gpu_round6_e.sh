#!/bin/bash
# Round 6, session E: off-policy acting on weight images on the environment's block; counters of the SAC update with
# the weight images and with the float32 passes.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
echo "== drop-in trajectories + learning (uneven torso)"
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider -k "drop_in" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_learning.py -q --timeout 600 -p no:cacheprovider -k "SAC or TD3 or DDPG" 2>&1 | tail -6
echo "== off-policy loops, block path"
timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06e_offpolicy_loops_block.txt
echo "== off-policy loops, staged copies (TONIC_AMD_Q_BLOCK=0)"
TONIC_AMD_Q_BLOCK=0 timeout 600 python scripts/offpolicy_loops.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06e_offpolicy_loops_staged.txt
echo "== counters, weight images"
bash scripts/gpu_pmc_sac.sh > gpurun_out/r06e_pmc_images.log 2>&1
for p in a b c; do cp gpurun_out/pmc_sac_$p.csv gpurun_out/r06e_pmc_sac_images_$p.csv; done
echo "== counters, float32 passes"
TONIC_AMD_TUNING=q_images=0 bash scripts/gpu_pmc_sac.sh > gpurun_out/r06e_pmc_f32.log 2>&1
for p in a b c; do cp gpurun_out/pmc_sac_$p.csv gpurun_out/r06e_pmc_sac_f32_$p.csv; done
tail -3 gpurun_out/r06e_pmc_images.log gpurun_out/r06e_pmc_f32.log
