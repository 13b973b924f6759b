#!/bin/bash
# rocprofv3 kernel trace of a few off-policy learner updates: gpu_profile_offpolicy.sh <kind> [B]
set -u
KIND=${1:-mpo}; B=${2:-100}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_op && mkdir -p /tmp/prof_op
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_op -o op -- python $REPO/scripts/offpolicy_update.py $KIND eager $B > $REPO/gpurun_out/prof_$KIND.log 2>&1
STATS=$(find /tmp/prof_op -name "*kernel_stats.csv" | head -1)
cp "$STATS" $REPO/gpurun_out/prof_${KIND}_kernel_stats.csv
tail -1 $REPO/gpurun_out/prof_$KIND.log
python3 - "$STATS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'].split('(')[0][-52:]:52s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us {r['Percentage']:>6s}%")
PY
