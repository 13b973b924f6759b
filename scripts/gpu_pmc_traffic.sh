#!/bin/bash
# HBM traffic of the GAE scan and the fused grad kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (they do not fit one pass on gfx950), counters + kernel-trace only.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/scripts/pmc_traffic.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $c failed"; tail -5 /tmp/pmc_$c.log; continue; fi
  python3 - "$f" $c $REPO/gpurun_out/pmc_traffic_$c.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0]
    if 'tonic::' not in k: continue
    agg[k].append(float(r['Counter_Value']))
with open(sys.argv[3], 'w') as out:
    out.write('kernel,counter,launches,mean_value_per_launch\n')
    for k, v in agg.items():
        out.write(f'"{k}",{sys.argv[2]},{len(v)},{sum(v)/len(v):.1f}\n')
        print(f'{k[-70:]:72s} {sys.argv[2]} n={len(v)} mean={sum(v)/len(v):.1f}')
PY
done
