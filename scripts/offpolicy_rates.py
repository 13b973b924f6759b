"""Developer probe: learner updates/s of the off-policy agents (bench.offpolicy_rates without the
CPU leg) — SAC cfg 3, TD3 cfg-4 share at B = 100 / 1024, optionally D4PG / MPO."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out = {'sac': bench.offpolicy_rates('sac', cpu=False)}
for b in (100, 1024):
    out[f'td3_B{b}'] = bench.offpolicy_rates('td3', 67, 21, b, workers=64, cpu=False)
if 'all' in sys.argv[1:]:
    for kind in ('d4pg', 'mpo'):
        out[kind] = bench.offpolicy_rates(kind, 67, 21, 100, workers=64, cpu=False)
for key, value in out.items():
    print(key, json.dumps({k: v for k, v in value.items() if k != 'workload'}))
