"""Developer probe: the collect loop of the bench workload (256 workers, O = 17, A = 6) with 1, 2 and 4
polls of the command word in flight per workgroup of the resident kernel (TONIC_AMD_COLLECTOR_POLLS),
one agent alive at a time, variants alternating on one box: microseconds per environment step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variants = sys.argv[2].split(',') if len(sys.argv) > 2 else ['1:1', '2:1', '4:1']      # depth:pause
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        os.environ['TONIC_AMD_COLLECTOR_POLLS'], os.environ['TONIC_AMD_COLLECTOR_POLL_SLEEP'] = v.split(':')
        agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
        loop.run(bench.T - agent.replay.index)
        torch.cuda.synchronize()
        for _ in range(3):
            loop.run(64)
            t0 = time.perf_counter()
            loop.run(1024)
            times[v].append((time.perf_counter() - t0) / 1024 * 1e6)
        agent.close()
        del agent, loop, rollout
for v in variants:
    print('polls in flight : pause', v, ': us per environment step', ' '.join(f'{t:.2f}' for t in times[v]),
          '| median', round(float(np.median(times[v])), 2))
