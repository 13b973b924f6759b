"""Prints a table of per-kernel register / scratch / occupancy figures for one .hip file
(`hipcc -Rpass-analysis=kernel-resource-usage`).  Developer tool, not part of the product."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
       '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass-analysis', line)
    if not m:
        continue
    key, _, val = m.group(1).partition(':')
    if key.strip() == 'Function Name':
        cur = {'name': val.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[key.strip()] = val.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = name.replace('tonic::', '').replace('(MlpArgs)', '').replace('void ', '')
    print(f"{name[:58]:58s} vgpr={r.get('VGPRs'):>4s} agpr={r.get('AGPRs'):>3s} "
          f"spill={r.get('VGPRs Spill'):>4s} scratch={r.get('ScratchSize [bytes/lane]'):>5s} "
          f"occ={r.get('Occupancy [waves/SIMD]')} sgpr={r.get('TotalSGPRs')}")
