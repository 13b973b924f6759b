cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python scripts/step_timeline.py run > /tmp/tl/run.log 2>&1
python scripts/step_timeline.py /tmp/tl > gpurun_out/r05_step_timeline.txt 2>&1
