"""bench.py's strong leg under two ranks sharing one GPU: where does the time go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tonic_amd import parallel
rank, world = parallel.init_from_env()
import torch
keep = []
for W in (256, 128):
    agent = bench.build_agent(seed=0)
    keep.append(agent)
    loop = bench.HostLoop(agent, W, seed=1 + rank)
    keep.append(loop)
    loop.run(100)
    t0 = time.perf_counter(); loop.run(1000); dt = time.perf_counter() - t0
    print('rank', rank, 'workers', W, 'us per env step', round(dt / 1000 * 1e6, 2), flush=True)
    loop.run(bench.T - agent.replay.index - 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.run(1)                      # the step that carries the learner update
    torch.cuda.synchronize()
    print('rank', rank, 'workers', W, 'update ms', round((time.perf_counter() - t0) * 1e3, 1), flush=True)
torch.distributed.barrier()
