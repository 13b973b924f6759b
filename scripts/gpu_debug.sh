cd ${GRAFT_REPO_ROOT:-/root/repo}
{
python scripts/gpu_debug.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_collector.py tests/test_gpu_learning.py tests/test_reference_dropin.py -m gpu -q -x 2>&1 | tail -5
} > gpurun_out/r05_debug.txt 2>&1
