"""gpurun_out/pmc_traffic_{FETCH,WRITE}_SIZE.csv (scripts/gpu_pmc_traffic.sh) -> profiles/rNN_traffic.json
with the gfx950 read correction (FETCH_SIZE counts 64 B per 128-byte request: read bytes =
FETCH_SIZE [KB] x 1024 x 2; MI355X_MICROARCH.md, HBM section) and the algorithmic bytes beside them."""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
N, T, W, T2, W2, O, A = 4096 * 256, 4096, 65536, 4096, 256, 17, 6
Pa, Pc = 64 * O + 64 + 4096 + 64 + A + 64 * A + A, 64 * O + 64 + 4096 + 64 + 64 + 1
ALGORITHMIC = {      # bytes per launch
    'gae_scan_kernel': ('gae_scan_kernel@T=4096,W=65536', 28 * T * W),
    'gae_stream16_kernel': ('gae_stream16_kernel@T=4096,W=256', 28 * T2 * W2),
    'gae_onepass_kernel': ('gae_onepass_kernel@T=4096,W=256', 28 * T2 * W2),
    'actor': ('mlp64_grad16_kernel<actor>@N=1048576', N * (O + A + 2) * 4 + 256 * (Pa + 8 + 55) // 64 * 64 * 4),
    'critic': ('mlp64_grad16_kernel<critic>@N=1048576', N * (O + 1) * 4 + 256 * (Pc + 8 + 55) // 64 * 64 * 4),
    'values': ('mlp64_grad16_kernel<values>@N=1048576', N * (O + 1) * 4),
}


def short(name):
    name = name.replace('tonic::', '').replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'mlp64_grad16_kernel<\d+, \d+, \d+, \d+, (true|false), (true|false), \d+, (true|false)(?:, (true|false))?>', name)
    if m:
        return 'values' if m.group(4) == 'true' else 'actor' if m.group(1) == 'true' else 'critic'
    if name.startswith('gae_onepass_kernel'):
        return 'gae_onepass_kernel'
    return name.split('<')[0] if name.startswith(('gae_', 'reduce_')) and '<' not in name else name


def table(counter):
    path = os.path.join(ROOT, 'gpurun_out', f'pmc_traffic_{counter}.csv')
    shutil.copy(path, os.path.join(ROOT, 'profiles', f'{rnd}_pmc_traffic_{counter}.csv'))
    return {short(r['kernel']): float(r['mean_value_per_launch']) for r in csv.DictReader(open(path))}


fetch, write = table('FETCH_SIZE'), table('WRITE_SIZE')
kernels = {}
for key in fetch:
    label, algorithmic = ALGORITHMIC.get(key, (key, None))
    kernels[label] = dict(read_bytes=int(fetch[key] * 1024 * 2), write_bytes=int(write.get(key, 0.0) * 1024))
    if algorithmic:
        kernels[label]['algorithmic_bytes'] = int(algorithmic)
out = dict(
    source=f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/gpu_pmc_traffic.sh, workload '
           f'scripts/pmc_traffic.py), MI355X, round {rnd[1:]} (raw: {rnd}_pmc_traffic_{{FETCH,WRITE}}_SIZE.csv)',
    correction='gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes = FETCH_SIZE(KB) * 1024 * 2; '
               'WRITE_SIZE(KB) * 1024 as is (MI355X_MICROARCH.md, HBM section); calibrated on the GAE scan at '
               'W = 65 536, whose byte count is known exactly',
    kernels=kernels)
json.dump(out, open(os.path.join(ROOT, 'profiles', f'{rnd}_traffic.json'), 'w'), indent=1)
for k, v in kernels.items():
    total = v['read_bytes'] + v['write_bytes']
    ratio = f"{total / v['algorithmic_bytes']:.2f}x algorithmic" if 'algorithmic_bytes' in v else ''
    print(f'{k[:60]:60s} read {v["read_bytes"] / 1e6:10.2f} MB  written {v["write_bytes"] / 1e6:9.2f} MB  {ratio}')
