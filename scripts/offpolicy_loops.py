"""Developer probe: BASELINE configs 3 / 4 on their own metric — env steps/s + learner updates/s of the whole
agent.step -> environment.step -> agent.update loop (bench.offpolicy_loop)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print('sac', json.dumps(bench.offpolicy_loop('sac', 111, 8, 1024, workers=1, loop_iterations=2000)))
print('td3', json.dumps(bench.offpolicy_loop('td3', 67, 21, 100, workers=64, loop_iterations=300)))
