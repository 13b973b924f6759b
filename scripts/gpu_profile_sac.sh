#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sac
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sac -o sac -- python $REPO/scripts/sac_update.py $MODE > $REPO/gpurun_out/sac_prof.log 2>&1
cp /tmp/prof_sac/sac_kernel_stats.csv $REPO/gpurun_out/sac_kernel_stats.csv
head -6 /tmp/prof_sac/sac_kernel_stats.csv | cut -c1-150
python $REPO/scripts/kernel_sequence.py /tmp/prof_sac/sac_kernel_trace.csv 17
grep "ms per" $REPO/gpurun_out/sac_prof.log
