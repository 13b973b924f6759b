#!/bin/bash
# Round 6, closing session: the whole -m gpu suite, smoke, the driver's bench command.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd $REPO
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rf 2>&1 | tail -25 | tee gpurun_out/r06_gpu_tests.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (the driver's command)"
timeout 900 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench_line.json; echo; tail -2 gpurun_out/r06_bench.err
echo "== step probe"
timeout 300 python scripts/offpolicy_step_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_offpolicy_step_probe.txt
