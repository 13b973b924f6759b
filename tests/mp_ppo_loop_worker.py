"""Worker of tests/test_gpu_multirank.py: one rank of a world_size-N PPO job driven through the host
loop (agent.step / environment.step / agent.update, pinned-host collector) for three rollouts +
updates: each rank steps its own workers, the updates exchange gradient sums.  Every rank saves
its parameters, the last update's logged rows and whether the critic's iterations ran under the
next rollout (TONIC_AMD_CRITIC_OVERLAP)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tonic_amd                      # noqa: E402
import tonic_amd.torch                # noqa: E402
from tonic_amd import environments, parallel    # noqa: E402


def run(out_path, T=48, W=32, O=17, A=6, iterations=8):
    rank, world = parallel.init_from_env()
    env = environments.SyntheticBatch(W, O, A, max_episode_steps=20, pool=7)
    env.initialize(seed=3 + rank)
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=T, batch_iterations=iterations))
    agent.initialize(env.observation_space, env.action_space, seed=11)
    observations = env.start()
    overlapped = 0
    for t in range(3 * T + 5):
        actions = agent.step(observations, t * W * world)
        observations, infos = env.step(actions)
        agent.update(**infos, steps=t * W * world)
        if (t + 1) % T == 0:
            overlapped += getattr(agent, '_critic_pending', None) is not None
    torch.cuda.synchronize()
    rows = np.array(agent.last_infos)
    state = {k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
    np.savez(out_path + f'.rank{rank}.npz', infos=rows, overlapped=np.array([overlapped]),
             exchange=np.array(str((parallel.allreduce_choice() or {}).get('kind', ''))), **state)
    agent.close()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1])
