set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/grad_variant_timing.py 1,4 120 2>&1 | tail -3 | tee gpurun_out/f16_timing.log
timeout 600 python -m pytest -q --timeout 300 -p no:cacheprovider tests/test_gpu_parity.py -k "golden_batch or clipping or full_size or reproducible or extreme or fp32_class or shape_bucket or whole or iterations or value" -s 2>&1 | grep -E "passed|failed|relative error|Error|assert" | tail -12 | tee gpurun_out/f16_tests2.log
