"""Host-loop cost of a second agent built after a first one in the same process (bench's strong leg)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
order = [int(x) for x in sys.argv[1:]] or [256, 128]
for W in order:
    agent = bench.build_agent(seed=0)
    loop = bench.HostLoop(agent, W, seed=1)
    loop.run(100)
    t0 = time.perf_counter(); loop.run(1000); dt = time.perf_counter() - t0
    print('workers', W, 'us per env step', round(dt / 1000 * 1e6, 2), flush=True)
