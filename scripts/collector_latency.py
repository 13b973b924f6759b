"""Where one host-in-the-loop environment step goes (W=256 HalfCheetah shapes unless told
otherwise): the raw C round trip of the collector (launch -> actions visible) for both
transports, then the drop-in agent loop with the time spent in agent.step / env.step /
agent.update.  Prints one JSON object; used for profiles/r02_collector_latency.md."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def raw_round_trip(W, O, A, transport, steps=2000):
    import torch
    from tonic_amd.collector import Block, Collector
    block = Block(W, O, A)
    collector = Collector(block, transport)
    T = 64
    seg = {k: torch.zeros(T, W, *tail, device='cuda') for k, tail in (
        ('observations', (O,)), ('actions', (A,)), ('next_observations', (O,)), ('rewards', ()),
        ('resets', ()), ('terminations', ()), ('log_probs', ()))}
    sums = torch.zeros(2 * O, device='cuda')
    collector.bind_segment(seg, sums, T)
    n = 64 * O + 64 + 4096 + 64 + A + 64 * A + A
    flat = torch.randn(n, device='cuda') * 0.1
    torch.cuda.synchronize()
    collector.begin_rollout(flat)
    block.observations[:] = np.random.randn(W, O)
    block.eps[0][:] = np.random.randn(W, A)
    block.eps[1][:] = np.random.randn(W, A)
    lat = np.zeros(steps)
    launch = np.zeros(steps)
    for i in range(steps + 100):
        t0 = time.perf_counter()
        collector.ppo_step(i % T, i & 1, i % T > 0)
        t1 = time.perf_counter()
        collector.wait_actions()
        t2 = time.perf_counter()
        if i >= 100:
            lat[i - 100] = t2 - t0
            launch[i - 100] = t1 - t0
    collector.end_rollout(-1)
    collector.close()
    return dict(transport=transport, round_trip_us=dict(
        median=round(float(np.median(lat)) * 1e6, 2), p10=round(float(np.percentile(lat, 10)) * 1e6, 2),
        p90=round(float(np.percentile(lat, 90)) * 1e6, 2)),
        launch_call_us=round(float(np.median(launch)) * 1e6, 2))


def agent_loop(W, O, A, transport, steps=3000, pool=64):
    import torch
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import SyntheticBatch
    os.environ['TONIC_AMD_COLLECTOR_TRANSPORT'] = str(transport)
    env = SyntheticBatch(W, O, A, max_episode_steps=1000, pool=pool)
    env.initialize(seed=1)
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=steps + 200, batch_iterations=1))
    agent.initialize(env.observation_space, env.action_space, seed=0)
    observations = env.start()
    clock = time.perf_counter
    parts = np.zeros(3)
    for t in range(steps + 100):
        if t == 100:
            parts[:] = 0
            begin = clock()
        t0 = clock()
        actions = agent.step(observations, t * W)
        t1 = clock()
        observations, infos = env.step(actions)
        t2 = clock()
        agent.update(**infos, steps=t * W)
        t3 = clock()
        parts += (t1 - t0, t2 - t1, t3 - t2)
    total = clock() - begin
    agent._collector.end_rollout(agent.replay.index - 1)
    torch.cuda.synchronize()
    return dict(transport=transport, us_per_step=round(total / steps * 1e6, 2),
                agent_step_us=round(parts[0] / steps * 1e6, 2),
                env_step_us=round(parts[1] / steps * 1e6, 2),
                agent_update_us=round(parts[2] / steps * 1e6, 2),
                env_steps_per_sec=round(W * steps / total, 1))


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--workers', type=int, default=256)
    parser.add_argument('--obs', type=int, default=17)
    parser.add_argument('--act', type=int, default=6)
    args = parser.parse_args()
    out = dict(W=args.workers, O=args.obs, A=args.act, raw=[], agent=[])
    transports = [int(t) for t in os.environ.get('TRANSPORTS', '2,0,1').split(',')]
    for transport in transports:
        out['raw'].append(raw_round_trip(args.workers, args.obs, args.act, transport))
    if os.environ.get('RAW_ONLY') != '1':
        for transport in transports:
            out['agent'].append(agent_loop(args.workers, args.obs, args.act, transport))
    print(json.dumps(out))
