#!/bin/bash
# PMC passes over the off-policy kernels (scripts/sac_update.py).  Counters + kernel-trace only.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TA_[A-Z0-9_]+)\b" | sort -u > $REPO/gpurun_out/counters_tc.txt
wc -l $REPO/gpurun_out/counters_tc.txt
pass() {
  name=$1; shift
  rm -rf /tmp/pmcs_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs_$name -o p -- python $REPO/scripts/sac_update.py > /tmp/pmcs_$name.log 2>&1
  f=$(find /tmp/pmcs_$name -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $name failed"; tail -5 /tmp/pmcs_$name.log; return; fi
  python3 - "$f" $REPO/gpurun_out/pmc_sac_$name.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:] + '/g' + r['Grid_Size']
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(sys.argv[2], 'w') as out:
    out.write('kernel,counter,mean_per_launch,launches\n')
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            out.write(f'{k},{c},{sum(v)/len(v):.1f},{len(v)}\n')
            if 'mlp_' in k or 'gemm16' in k:
                print(f'{k:50s} {c:34s} {sum(v)/len(v):14.0f}  (n={len(v)})')
PY
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD
pass b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass c GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum
