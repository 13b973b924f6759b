"""Where the host-in-the-loop environment step goes INSIDE agent.step / agent.update (cfg 2 shapes, the bench's
HostLoop): time blocked in tonic_collector_wait_actions (the GPU's round trip minus what the host overlapped),
in the arm / claim calls and in the noise copy, per environment step.  usage: host_loop_probe.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
import torch                                     # noqa: E402
from tonic_amd.utils import logger               # noqa: E402
logger.get_current_logger().store = lambda *a, **k: None
agent = bench.build_agent(seed=0)
loop = bench.HostLoop(agent, bench.W, seed=1)
loop.run(bench.T)                                # one whole rollout + update: everything bound and warm
agent.settle()
torch.cuda.synchronize()
loop.run(64)
collector, noise = agent._collector, agent._noise
clock = time.perf_counter
spent = dict(wait=0.0, arm=0.0, claim=0.0, take=0.0)


def timed(name, fn):
    def call(*args):
        t0 = clock()
        out = fn(*args)
        spent[name] += clock() - t0
        return out
    return call


collector._wait = timed('wait', collector._wait)
collector._arm = timed('arm', collector._arm)
collector._claim = timed('claim', collector._claim)
noise.take = timed('take', noise.take)
base = loop.breakdown(steps)
out = dict(base, inside_us={k: round(v / steps * 1e6, 2) for k, v in spent.items()},
           note='the four inner timers add ~0.15 us each to the totals they sit in')
print(json.dumps(out))
