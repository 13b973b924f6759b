"""Developer probe: where an off-policy loop iteration goes on the host (SAC, cfg-3 shapes, one worker; TD3 share, 64
workers): the pieces of agent.step / agent.update on the environment's block, timed one by one."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tonic_amd import _lib  # noqa: E402
from tonic_amd.environments import SyntheticBatch  # noqa: E402


def probe(kind, o_dim, a_dim, batch, workers, count=2000):
    agent, replay = bench.build_offpolicy(kind, o_dim, a_dim, batch, workers, 50, 1000000)
    env = SyntheticBatch(workers, o_dim, a_dim, max_episode_steps=1000, pool=64)
    env.initialize(seed=3)
    observations = env.start()
    steps = 100000
    replay.last_steps = 10 ** 9                      # no learner update: the loop's own cost
    clock = time.perf_counter
    for _ in range(50):
        actions = agent.step(observations, steps)
        observations, infos = env.step(actions)
        agent.update(**infos, steps=steps)
    torch.cuda.synchronize()
    out = {}
    state = agent._block_of(observations, agent.policy_kind)
    block, collector = state['block'], state['collector']
    kind_code, stochastic = agent.policy_kind, kind == 'sac'
    parts = dict(block_of=0.0, noise=0.0, launch=0.0, wait=0.0, copy_out=0.0)
    for _ in range(count):
        t0 = clock()
        agent._block_of(observations, kind_code)
        t1 = clock()
        if stochastic:
            np.copyto(block.eps[0], agent._randn(workers, a_dim).numpy())
        t2 = clock()
        _lib.check(agent._q_act(
            collector.handle, _lib.ptr(agent.model.flat_actor.flat), _lib.ptr(agent._actor_images), 0, kind_code,
            agent.hidden, 0 if stochastic else -1, _lib.ptr(state['rows'][0]), None, _lib.ptr(state['workspace']),
            state['workspace'].numel(), _lib.current_stream()), 'q_act')
        t3 = clock()
        collector.wait_actions()
        t4 = clock()
        block.eps[1].copy()
        t5 = clock()
        for key, dt in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            parts[key] += dt
    out['act_pieces_us'] = {k: round(v / count * 1e6, 2) for k, v in parts.items()}
    whole = np.zeros(3)
    for _ in range(count):
        t0 = clock()
        actions = agent.step(observations, steps)
        t1 = clock()
        observations, infos = env.step(actions)
        t2 = clock()
        agent.update(**infos, steps=steps)
        whole += (t1 - t0, t2 - t1, clock() - t2)
    out['loop_us'] = dict(zip(('agent_step', 'env_step', 'agent_update'), np.round(whole / count * 1e6, 2).tolist()))
    # the store launch alone
    t0 = clock()
    for _ in range(count):
        replay.store(normalizer=agent.model.observation_normalizer, observations=state['rows'][0], **state['fields'])
    out['store_call_us'] = round((clock() - t0) / count * 1e6, 2)
    torch.cuda.synchronize()
    return out


print('sac', json.dumps(probe('sac', 111, 8, 1024, 1)))
print('td3', json.dumps(probe('td3', 67, 21, 100, 64)))
