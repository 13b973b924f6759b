"""Runs a few SAC learner updates at cfg-3 shapes (target of rocprofv3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tonic_amd, tonic_amd.torch as tt
from tonic_amd.environments import Box
O, A, B, iters, rows = 111, 8, 1024, 50, 100000
replay = tonic_amd.replays.Buffer(size=rows, batch_iterations=iters, batch_size=B)
agent = tt.agents.SAC(replay=replay)
agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
replay._allocate(1, O, A)
for k, b in replay.buffers.items():
    b.copy_(torch.randn(b.shape, device='cuda') * (0.0 if k in ('resets', 'terminations') else 1.0))
replay.buffers['discounts'].fill_(0.99)
replay.size = rows
graph = len(sys.argv) > 1 and sys.argv[1] == 'graph'
for _ in range(3):
    agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=graph)
torch.cuda.synchronize()
print('ms per iteration', (time.perf_counter() - t0) / 3 / iters * 1e3)
