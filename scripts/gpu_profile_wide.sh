#!/bin/bash
# rocprofv3 kernel trace of the layer-by-layer PPO path (scripts/wide_timing.py)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w && mkdir -p /tmp/prof_w
WIDE_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o w -- python $REPO/scripts/wide_timing.py > $REPO/gpurun_out/prof_wide.log 2>&1
STATS=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1)
cp "$STATS" $REPO/gpurun_out/prof_wide_kernel_stats.csv
grep "O=" $REPO/gpurun_out/prof_wide.log
python3 - "$STATS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s}%")
PY
