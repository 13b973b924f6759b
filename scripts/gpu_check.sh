#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo | grep -E "Marketing|gfx" | head -4
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
