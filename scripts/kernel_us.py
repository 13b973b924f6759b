"""profiles/rNN_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py`) ->
profiles/rNN_kernel_us.json: average duration per kernel in us under short names, which bench.py
quotes as `rocprof_us` next to its live HIP-event figures.  usage: kernel_us.py [round]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', f'{rnd}_bench_kernel_stats.csv'))))
table = {}
for r in rows:
    name = r['Name'].replace('tonic::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    if not ('tonic::' in r['Name']):
        continue
    short = name.split('(')[0]
    m = re.match(r'mlp64_grad16_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+), '
                 r'(true|false), (true|false)>', short)
    if m:
        role = 'values' if m.group(9) == 'true' else 'actor' if m.group(5) == 'true' else 'critic'
        # the metric's shapes (O = 17: KS1 = 5) get the plain key, other buckets carry theirs
        short = (f'mlp64_grad16_kernel<{role}>' if m.group(1) == '5'
                 else f'mlp64_grad16_kernel<{role}, KS1={m.group(1)}>')
    entry = dict(us=round(float(r['AverageNs']) / 1e3, 2), calls=int(r['Calls']))
    # (variants that share a short name — the grad kernels' bf16x3 / fp32 builds, measured by the
    #  roofline legs a few times each — : the one the update runs, i.e. the most calls, keeps the name)
    if short not in table or table[short]['calls'] < entry['calls']:
        table[short] = entry
out = dict(source=f'profiles/{rnd}_bench_kernel_stats.csv',
           command='rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1',
           kernels={k: v['us'] for k, v in table.items()}, calls={k: v['calls'] for k, v in table.items()})
json.dump(out, open(os.path.join(ROOT, 'profiles', f'{rnd}_kernel_us.json'), 'w'), indent=1)
for k, v in sorted(table.items(), key=lambda kv: -kv[1]['us'] * kv[1]['calls'])[:14]:
    print(f"{k[:70]:70s} {v['us']:10.2f} us x {v['calls']}")
