"""Worker of tests/test_gpu_multirank.py: one rank of a world_size-N TD3 / SAC learner update.
Each rank owns a contiguous shard of the worker axis of a common synthetic Buffer; every rank
draws the same GLOBAL index and noise streams (same seed) and keeps its own samples."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tonic_amd                      # noqa: E402
import tonic_amd.torch                # noqa: E402
from tonic_amd import parallel        # noqa: E402
from tonic_amd.environments import Box  # noqa: E402


def run(kind, out_path, rows=40, W=8, O=11, A=3, iterations=6, batch=48):
    rank, world = parallel.init_from_env()
    cls = dict(sac=tonic_amd.torch.agents.SAC, td3=tonic_amd.torch.agents.TD3,
               d4pg=tonic_amd.torch.agents.D4PG, mpo=tonic_amd.torch.agents.MPO)[kind]
    agent = cls(replay=tonic_amd.replays.Buffer(size=rows * W, batch_iterations=iterations,
                                                batch_size=batch, steps_before_batches=0,
                                                steps_between_batches=1))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=7)
    rng = np.random.RandomState(321)
    lo, hi = parallel.shard_bounds(W)
    norm = agent.model.observation_normalizer
    for _ in range(rows - 3):                       # partially filled circular buffer
        row = dict(observations=rng.normal(size=(W, O)), actions=rng.uniform(-1, 1, size=(W, A)),
                   next_observations=rng.normal(size=(W, O)), rewards=rng.normal(size=W),
                   resets=rng.uniform(size=W) < 0.1, terminations=rng.uniform(size=W) < 0.05)
        row = {k: torch.as_tensor(np.ascontiguousarray(np.asarray(v, np.float32)[lo:hi])).cuda()
               for k, v in row.items()}
        agent.replay.store(normalizer=norm, **row)
    agent._update(steps=rows * W)
    torch.cuda.synchronize()
    if rank == 0:
        state = {k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
        if kind == 'mpo':           # the dual variables and the logged losses of the actor step
            state['duals'] = agent.actor_updater.duals.cpu().numpy()
            state['actor_infos'] = agent.last_actor_infos
        np.savez(out_path, infos=agent.last_infos, **state)
    # Buffer.get (buffers.py:81-91) across ranks: tiny batches so that some rank draws nothing
    replay = agent.replay
    replay.np_random = np.random.RandomState(99)
    replay.batch_size = 3
    parts = list(replay.get('observations', 'rewards', steps=rows * W))
    np.savez(out_path + f'.get{rank}.npz',
             **{f'rewards{i}': b['rewards'].cpu().numpy() for i, b in enumerate(parts)},
             **{f'observations{i}': b['observations'].cpu().numpy() for i, b in enumerate(parts)})
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1], sys.argv[2])
