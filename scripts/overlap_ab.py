"""Developer probe: the bench workload's step (T = 4096 environment steps of 256 workers + one learner
update) with and without the critic's iterations running under the next rollout
(TONIC_AMD_CRITIC_OVERLAP), one agent alive at a time, alternating on one box: ms per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
times = {'0': [], '1': []}
for r in range(rounds):
    for v in ('0', '1'):
        os.environ['TONIC_AMD_CRITIC_OVERLAP'] = v
        agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
        loop.run(bench.T - agent.replay.index)
        torch.cuda.synchronize()
        for _ in range(2):
            # (like bench.py: several steps between two device syncs — a sync after every step would
            #  wait for the critic's chain each time and measure no overlap at all)
            t0 = time.perf_counter()
            for _ in range(4):
                loop.run(bench.T)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) * 1e3 / 4)
        print('overlap', v, loop.breakdown(), flush=True)
        agent.close()
        del agent, loop, rollout
for v in times:
    print('TONIC_AMD_CRITIC_OVERLAP', v, ': ms per step', ' '.join(f'{t:.1f}' for t in times[v]),
          '| median', round(float(np.median(times[v])), 2), '->', round(bench.T * 256 / np.median(times[v]) / 1e3, 2), 'M env-steps/s')
