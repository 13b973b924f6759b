#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- python $REPO/bench.py --steps 2 --warmup 1 --no-extras > $REPO/gpurun_out/collect_prof.log 2>&1
head -8 /tmp/prof_c/c_kernel_stats.csv | cut -c1-150
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/tmp/prof_c/c_kernel_trace.csv')) if 'collect16' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
import statistics
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
g=[int(b['Start_Timestamp'])-int(a['End_Timestamp']) for a,b in zip(rows,rows[1:])]
g=[x for x in g if x<1e6]
print('n',len(rows),'dur median',statistics.median(d),'gap median',statistics.median(g),'gap mean',sum(g)/len(g))
PY
