"""How often agent.update issues the next step early in bench.py's host loop, and what the loop costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
agent = bench.build_agent(seed=0)
loop = bench.HostLoop(agent, bench.W, seed=1)
loop.run(50)
early = 0
obs, steps = loop.observations, loop.steps
for _ in range(200):
    actions = agent.step(obs, steps)
    obs, infos = loop.env.step(actions)
    agent.update(**infos, steps=steps)
    early += bool(agent._speculated)
    steps += bench.W
print('issued early', early, 'of 200; fed', agent._block_fed, 'eps_ahead', agent._eps_ahead)
loop.observations, loop.steps = obs, steps
t0 = time.perf_counter(); loop.run(2000); dt = time.perf_counter() - t0
print('us per env step', dt / 2000 * 1e6)
print(loop.breakdown(1000))
