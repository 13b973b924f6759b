"""Where a PPO step of the metric goes BETWEEN two rollouts, from the GPU's side.
  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -- python scripts/step_timeline.py run
  analyse:  python scripts/step_timeline.py DIR
`run` drives 5 steps of the bench's host loop (cfg 2 shapes); the analysis takes the last two rollouts of the
trace (resident collect kernels) and lists what ran between them: kernel groups with start / end relative to
the END of the first rollout's kernel, and the gaps."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == 'run':
    import bench
    import torch
    from tonic_amd.utils import logger
    logger.get_current_logger().store = lambda *a, **k: None
    agent = bench.build_agent(seed=0)
    loop = bench.HostLoop(agent, bench.W, seed=1)
    for _ in range(5):
        loop.run(bench.T)
    agent.settle()
    torch.cuda.synchronize()
    sys.exit(0)

path = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('tonic::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:58]
res = [i for i, r in enumerate(rows) if 'ppo_collect_resident_kernel' in r['Kernel_Name']]
# a rollout = resident launches less than 1 ms apart (the kernel parks and is started again at most a few times)
groups = []
for i in res:
    s0, e0 = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    if groups and s0 - groups[-1][1] < 1000000:
        groups[-1][1] = e0; groups[-1][2] += 1
    else:
        groups.append([s0, e0, 1])
print('rollouts (ms from the first):', [(round((g[0] - groups[0][0]) / 1e6, 2), round((g[1] - g[0]) / 1e6, 2), g[2]) for g in groups])
(a0, a1, _), (b0, b1, _) = groups[-2], groups[-1]
print(f'last full step: rollout {(a1 - a0) / 1e6:.3f} ms, then {(b0 - a1) / 1e6:.3f} ms until the next rollout starts')
agg = {}
for r in rows:
    s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'ppo_collect_resident_kernel' in r['Kernel_Name'] or e0 < a0 or s0 > b1:
        continue
    n = short(r['Kernel_Name'])
    g = agg.setdefault(n, [s0, e0, 0, 0])
    g[0] = min(g[0], s0); g[1] = max(g[1], e0); g[2] += 1; g[3] += e0 - s0
print('kernels between the START of that rollout and the END of the next (ms relative to the rollout\'s end):')
for n, (s0, e0, count, busy) in sorted(agg.items(), key=lambda kv: kv[1][0]):
    print(f'{(s0 - a1) / 1e6:9.3f} .. {(e0 - a1) / 1e6:9.3f} ms  x{count:<5d} busy {busy / 1e6:8.3f} ms  {n}')
