"""Developer probe: wall time of consecutive host-in-the-loop steps (T = 4096 environment steps of 256
workers + one learner update each) of the bench workload, one line per step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.run(bench.T)
    torch.cuda.synchronize(); print(i, round((time.perf_counter() - t0) * 1e3, 2))
