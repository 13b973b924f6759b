#!/bin/bash
# Round 3, call C: L2 warm-up touches (forward stamps, rates, kernel sequence).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in 0 1; do
echo "== l2_touch=$t: forward stamps B=1024"
TONIC_AMD_TUNING=l2_touch=$t,q_chain=0 timeout 300 python scripts/forward_stamps.py 1024 2>&1 | tail -9 | cut -c1-200
echo "== l2_touch=$t: rates"
TONIC_AMD_TUNING=l2_touch=$t timeout 300 python scripts/offpolicy_rates.py 2>&1 | grep -o '^[a-z0-9_B]* \|"hip_graph": {[^}]*}\|"us_per_iteration": [0-9.]*' | paste - - - | tee gpurun_out/rates_touch$t.log
done
echo "== rocprof sac (touch on)"
bash scripts/gpu_profile_sac.sh 2>&1 | tail -8
echo "== fused tests"
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q -x -k "fused or full_size" 2>&1 | tail -3
