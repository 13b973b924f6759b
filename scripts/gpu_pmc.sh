#!/bin/bash
# PMC passes over the fused grad kernels.  Counters only (no tracing domains besides kernel-trace).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE)\b" | sort -u > $REPO/gpurun_out/counters.txt
wc -l $REPO/gpurun_out/counters.txt
WAVES=${1:-4}
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/scripts/pmc_grad.py $WAVES 2 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $name failed"; tail -5 /tmp/pmc_$name.log; return; fi
  cp "$f" $REPO/gpurun_out/pmc_${name}_w$WAVES.csv
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0][-60:]
    if 'grad' not in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:34s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES
pass c GRBM_GUI_ACTIVE GRBM_COUNT
