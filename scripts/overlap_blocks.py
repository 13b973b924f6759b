"""Developer probe: microseconds per environment step over the 16 blocks of 256 steps of ONE rollout of
the bench workload, right after a learner update, with the critic's iterations running under the
rollout (TONIC_AMD_CRITIC_OVERLAP=1) and without; optional argv[1]: workgroups of the critic's launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import tonic_amd.torch as tt
if len(sys.argv) > 1:
    tt.agents.PPO.OVERLAP_BLOCKS = int(sys.argv[1])
for v in ('0', '1'):
    os.environ['TONIC_AMD_CRITIC_OVERLAP'] = v
    agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
    for rep in range(2):
        loop.run(bench.T - agent.replay.index)             # ends with a learner update
        t_critic = None
        blocks = []
        pending = getattr(agent, '_critic_pending', None)
        t_start = time.perf_counter()
        for b in range(15):
            t0 = time.perf_counter()
            loop.run(256)
            blocks.append((time.perf_counter() - t0) / 256 * 1e6)
            if pending is not None and t_critic is None and pending['done'].query():
                t_critic = (time.perf_counter() - t_start) * 1e3
        print('overlap', v, 'us per step by block of 256:', ' '.join(f'{x:.1f}' for x in blocks),
              '| critic chain done within', t_critic, 'ms', flush=True)
    agent.close()
    del agent, loop, rollout
