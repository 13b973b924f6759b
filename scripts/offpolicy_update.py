"""Runs a few learner updates of one off-policy agent at default network sizes (target of rocprofv3).
usage: offpolicy_update.py {sac|td3|ddpg|d4pg|mpo} [graph] [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tonic_amd, tonic_amd.torch as tt
from tonic_amd.environments import Box
kind = sys.argv[1] if len(sys.argv) > 1 else 'sac'
graph = len(sys.argv) > 2 and sys.argv[2] == 'graph'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 100
O, A, iters, rows = 67, 21, 50, 100000
replay = tonic_amd.replays.Buffer(size=rows, batch_iterations=iters, batch_size=B)
agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3, ddpg=tt.agents.DDPG, d4pg=tt.agents.D4PG,
             mpo=tt.agents.MPO)[kind](replay=replay)
agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
replay._allocate(1, O, A)
for k, b in replay.buffers.items():
    b.copy_(torch.randn(b.shape, device='cuda') * (0.0 if k in ('resets', 'terminations') else 1.0))
replay.buffers['discounts'].fill_(0.99)
replay.size = rows
for _ in range(3):
    agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=graph)
torch.cuda.synchronize()
print(kind, 'B', B, 'us per iteration', (time.perf_counter() - t0) / 3 / iters * 1e6)
