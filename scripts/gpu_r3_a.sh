#!/bin/bash
# Round 3, call A: the 2-rank bench test after agent.close(), mlp_waves = 8 (correctness + rates).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench ranks test"
timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -k "bench_starts or refuses" 2>&1 | tail -5
echo "== offpolicy tests, mlp_waves=8"
TONIC_AMD_TUNING=mlp_waves=8 timeout 900 python -m pytest tests/test_gpu_offpolicy.py -q 2>&1 | tail -8
echo "== rates, mlp_waves=4"
timeout 300 python scripts/offpolicy_rates.py 2>&1 | grep -v "^$" | cut -c1-400 | tee gpurun_out/rates_w4.log
echo "== rates, mlp_waves=8"
TONIC_AMD_TUNING=mlp_waves=8 timeout 300 python scripts/offpolicy_rates.py 2>&1 | cut -c1-400 | tee gpurun_out/rates_w8.log
echo "== rocprof sac, mlp_waves=8"
TONIC_AMD_TUNING=mlp_waves=8 bash scripts/gpu_profile_sac.sh 2>&1 | tail -24
