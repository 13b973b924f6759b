#!/bin/bash
# Round 3, call B: chained q-step launches (correctness + rates + kernel sequence).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== offpolicy tests"
timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q -x 2>&1 | tail -8
echo "== rates, q_chain=0"
TONIC_AMD_TUNING=q_chain=0 timeout 300 python scripts/offpolicy_rates.py 2>&1 | grep -o '^[a-z0-9_B]* \|"hip_graph": {[^}]*}\|"us_per_iteration": [0-9.]*' | paste - - - | tee gpurun_out/rates_chain0.log
echo "== rates, q_chain=1"
timeout 300 python scripts/offpolicy_rates.py 2>&1 | grep -o '^[a-z0-9_B]* \|"hip_graph": {[^}]*}\|"us_per_iteration": [0-9.]*' | paste - - - | tee gpurun_out/rates_chain1.log
echo "== rocprof sac"
bash scripts/gpu_profile_sac.sh 2>&1 | tail -16
