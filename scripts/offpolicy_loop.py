import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tonic_amd, tonic_amd.torch as tt
from tonic_amd.environments import Box, SyntheticBatch
for mode in ('mapped', 'copy'):
    os.environ['TONIC_AMD_STAGING'] = mode
    W, O, A = 1, 111, 8
    env = SyntheticBatch(W, O, A, pool=64, copy_outputs=True); env.initialize(0)
    agent = tt.agents.SAC(replay=tonic_amd.replays.Buffer(size=100000, batch_size=1024, steps_before_batches=10**9))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
    agent.exploration.start_steps = 0
    obs = env.start()
    parts = np.zeros(3); n = 2000
    for t in range(n + 100):
        if t == 100: parts[:] = 0
        t0 = time.perf_counter(); a = agent.step(obs, t); t1 = time.perf_counter()
        obs, infos = env.step(a); t2 = time.perf_counter()
        agent.update(**infos, steps=t); t3 = time.perf_counter()
        parts += (t1 - t0, t2 - t1, t3 - t2)
    torch.cuda.synchronize()
    print(mode, 'us per step: agent.step %.1f env %.1f agent.update %.1f' % tuple(parts / n * 1e6))
