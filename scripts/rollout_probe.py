"""Per-step wall time of six steps of the metric's host loop and the in-job rollout's microseconds per environment
step (agent._rollout_step_us of the rollout before); with TONIC_AMD_COLLECTOR_STAMPS=1 the collector prints its in-kernel
stamps when it is destroyed.  TONIC_AMD_CRITIC_OVERLAP=0: the critic's chain in front of the rollout instead of under it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from tonic_amd.utils import logger
logger.get_current_logger().store = lambda *a, **k: None
agent = bench.build_agent(seed=0)
loop = bench.HostLoop(agent, bench.W, seed=1)
out = []
for i in range(6):
    t0 = time.perf_counter()
    loop.run(bench.T)
    agent.settle() if os.environ.get('SETTLE') else None
    torch.cuda.synchronize() if os.environ.get('SETTLE') else None
    out.append((round((time.perf_counter() - t0) * 1e3, 2), round(getattr(agent, '_rollout_step_us', 0), 3)))
agent.settle(); torch.cuda.synchronize()
print(os.environ.get('TONIC_AMD_CRITIC_OVERLAP', '1'), os.environ.get('SETTLE', ''), out)
