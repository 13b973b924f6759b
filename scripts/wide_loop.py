"""Host-loop cost per environment step for shapes beyond the fused act kernel (staged path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tonic_amd, tonic_amd.torch
from tonic_amd.environments import SyntheticBatch
SHAPES = ((111, 8), (376, 17), (17, 6))
if os.environ.get("WIDE_ONE"):
    SHAPES = SHAPES[:1]
for O, A in SHAPES:
    W = 256
    env = SyntheticBatch(W, O, A, max_episode_steps=1000, pool=16)
    env.initialize(seed=1)
    agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=4096, batch_iterations=1))
    agent.initialize(env.observation_space, env.action_space, seed=0)
    obs = env.start()
    def run(n):
        global obs
        for t in range(n):
            actions = agent.step(obs, t * W)
            obs, infos = env.step(actions)
            agent.update(**infos, steps=t * W)
    run(100)
    t0 = time.perf_counter(); run(1000); dt = time.perf_counter() - t0
    print(f'O={O} A={A}: {dt / 1000 * 1e6:.1f} us per environment step of {W} workers', flush=True)
