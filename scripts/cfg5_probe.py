"""Developer probe: the host-loop breakdown (agent.step / environment.step / agent.update) of
BASELINE config 5's per-GPU share — AntBullet shapes, 1 280 workers."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.O, bench.A, bench.W = 28, 8, 1280
agent, loop, rollout, out = bench.measure_job(1280, 0, 1, 1, 1, True, device_too=False)
print({k: out[k] for k in ('value', 'ms_per_step')})
print(loop.breakdown(1024))
