"""Developer probe: the collect loop of the bench workload (256 workers, O = 17, A = 6) with and
without the environment issuing the armed step (TONIC_AMD_ARM), interleaved on one box:
microseconds per environment step over `rounds` x 2048 steps each, and the collector's stamps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

def job(arm):
    os.environ['TONIC_AMD_ARM'] = arm
    agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
    return agent, loop

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
jobs = {arm: job(arm) for arm in ('0', '1')}
times = {arm: [] for arm in jobs}
for r in range(rounds):
    for arm, (agent, loop) in jobs.items():
        loop.run(bench.T - agent.replay.index)             # finish the segment: a learner update
        torch.cuda.synchronize()
        loop.run(64)
        t0 = time.perf_counter()
        loop.run(2048)
        times[arm].append((time.perf_counter() - t0) / 2048 * 1e6)
for arm in jobs:
    print('TONIC_AMD_ARM', arm, 'us per environment step:', ' '.join(f'{t:.2f}' for t in times[arm]),
          '| median', round(float(np.median(times[arm])), 2),
          '| issued by the environment', jobs[arm][0].steps_issued_by_environment)
for arm, (agent, loop) in jobs.items():
    agent.close()
