#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gae
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gae -o gae -- python $REPO/scripts/gae_exact.py 2>&1 | grep "us per call"
python $REPO/scripts/kernel_sequence.py /tmp/prof_gae/gae_kernel_trace.csv 6
