#!/bin/bash
# rocprofv3 kernel trace of the default bench command; summary -> gpurun_out/prof_*.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 > $REPO/gpurun_out/bench_prof.log 2>&1
find /tmp/prof -type f | head -20
STATS=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$STATS" $REPO/gpurun_out/prof_kernel_stats.csv
gzip -c $(find /tmp/prof -name "*kernel_trace.csv" | head -1) > $REPO/gpurun_out/prof_kernel_trace.csv.gz
head -12 "$STATS"
tail -2 $REPO/gpurun_out/bench_prof.log | cut -c1-600
