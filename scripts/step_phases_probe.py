"""Where a step of the metric goes on the HOST's clock (no profiler): per step the rollout's wall time, the last
agent.update call (= the learner: end_rollout, evaluate, scan, 80 actor iterations enqueued + read back, the critic's
chain handed to its stream) and the first agent.step of the next rollout (begin_rollout + the resident kernel's
launch + the first round trip)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from tonic_amd.utils import logger  # noqa: E402

logger.get_current_logger().store = lambda *a, **k: None
agent = bench.build_agent(seed=0)
loop = bench.HostLoop(agent, bench.W, seed=1)
env, W, T = loop.env, bench.W, bench.T
observations, steps = loop.observations, 0
clock = time.perf_counter
rows = []
for step in range(7):
    t0 = clock()
    first = None
    for t in range(T):
        a0 = clock()
        actions = agent.step(observations, steps)
        if t == 0:
            first = clock() - a0
        observations, infos = env.step(actions)
        if t == T - 1:
            t1 = clock()
        agent.update(**infos, steps=steps)
        steps += W
    t2 = clock()
    rows.append(dict(step_ms=round((t2 - t0) * 1e3, 2), rollout_ms=round((t1 - t0) * 1e3, 2),
                     learner_call_ms=round((t2 - t1) * 1e3, 2), first_agent_step_ms=round(first * 1e3, 3),
                     actor_chain_ms=round(getattr(agent, 'actor_chain_ms', 0) or 0, 2)))
agent.settle()
torch.cuda.synchronize()
for r in rows:
    print(r)
