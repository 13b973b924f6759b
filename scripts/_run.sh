set -u
export TMPDIR=/tmp
timeout 400 python -m pytest -q --timeout 180 -p no:cacheprovider tests/test_gpu_learning.py -k "PPO" 2>&1 | tail -4
bash scripts/gpu_round4_profiles.sh
