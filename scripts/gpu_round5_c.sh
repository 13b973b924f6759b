#!/bin/bash
# Round 5: the whole GPU tier + the driver's bench command on the build with the push transport.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05c_tests.log
tail -6 gpurun_out/r05c_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/r05c_bench.err | tail -1 > gpurun_out/r05c_bench_line.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r05c_bench_line.json'))
print({k:b.get(k) for k in ('value','ms_per_step','collect_ms','host_loop','cfg5_share')})
PY
