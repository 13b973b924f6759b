cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_wl && mkdir -p /tmp/prof_wl
WIDE_ONE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wl -o w -- python $GRAFT_REPO_ROOT/scripts/wide_loop.py > /tmp/wl.log 2>&1
grep "O=" /tmp/wl.log
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/prof_wl/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'].split('(')[0][-46:], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
