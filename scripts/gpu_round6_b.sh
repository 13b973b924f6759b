#!/bin/bash
# Round 6, session B: the off-policy passes on fp16x2 weight images — parity tests, then rates with the images on / off.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== off-policy tests"
timeout 1500 python -m pytest tests/test_gpu_offpolicy.py -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -40 | tee gpurun_out/r06b_offpolicy_tests.log
echo "== the two tests that failed in session A"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "non_finite" -p no:cacheprovider 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_offpolicy.py -q -k "1024_units" -p no:cacheprovider 2>&1 | grep -E "Mismatch|Max abs|Max rel|assert|Error|passed|failed" | head -20
echo "== rates, images on"
timeout 600 python scripts/offpolicy_rates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_rates_images.txt
echo "== rates, images off (TONIC_AMD_TUNING=q_images=0)"
TONIC_AMD_TUNING=q_images=0 timeout 600 python scripts/offpolicy_rates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06b_rates_f32.txt
