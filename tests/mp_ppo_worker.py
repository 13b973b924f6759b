"""Worker of tests/test_gpu_multirank.py: one rank of a world_size-N PPO update (run with
RANK / WORLD_SIZE / MASTER_* set).  Each rank owns a contiguous shard of the worker axis of a
common synthetic Segment; rank 0 saves the updated parameters and statistics."""
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


def make_segment(T, W, O, A):
    rng = np.random.RandomState(123)
    seg = dict(observations=rng.normal(size=(T, W, O)), actions=np.clip(rng.normal(size=(T, W, A)), -1, 1),
               next_observations=rng.normal(size=(T, W, O)), rewards=rng.normal(size=(T, W)),
               resets=rng.uniform(size=(T, W)) < 0.1, terminations=rng.uniform(size=(T, W)) < 0.05,
               log_probs=rng.normal(size=(T, W)) * 0.1 - 6)
    return {k: np.asarray(v, np.float32) for k, v in seg.items()}


def run(out_path, T=12, W=16, O=17, A=6, iterations=4):
    rank, world = parallel.init_from_env()
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=T, batch_iterations=iterations))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=11)
    seg = make_segment(T, W, O, A)
    lo, hi = parallel.shard_bounds(W)
    norm = agent.model.observation_normalizer
    for t in range(T):
        row = {k: torch.as_tensor(np.ascontiguousarray(v[t, lo:hi])).cuda() for k, v in seg.items()}
        agent.replay.store(normalizer=norm, **row)
    agent._update()
    torch.cuda.synchronize()
    if rank == 0:
        state = {k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
        np.savez(out_path, infos=agent.last_infos, adv_stats=agent.replay.adv_stats.cpu().numpy(),
                 **state)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1])
