"""Times the fused actor / critic grad kernels at the benchmark size (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
import torch                                    # noqa: E402
from tonic_amd import _lib, replays             # noqa: E402
from tonic_amd.rollout import DeviceRollout     # noqa: E402

agent = bench.build_agent(seed=0)
rollout = DeviceRollout(agent, bench.W, bench.T, seed=1)
rollout.collect(capture=False)
agent._update()
lib, p = _lib.load(), _lib.ptr
replay, actor, critic = agent.replay, agent.actor_updater, agent.critic_updater
b = replay.buffers
n = bench.T * bench.W
obs, act, adv, logp, ret = (replays.flatten_batch(b[k]) for k in
                            ('observations', 'actions', 'advantages', 'log_probs', 'returns'))
mean, std = critic.norm_tensors()


def actor_grad():
    actor.enqueue_grad(obs, act, adv, replay.adv_stats, logp)


def critic_grad():
    critic.enqueue_grad(obs, ret)


for _ in range(3):
    ms_a, ms_c = bench.time_events(actor_grad, 20), bench.time_events(critic_grad, 20)
    print(f'actor {ms_a * 1e3:.1f} us ({bench.ACTOR_FLOP_PER_SAMPLE * n / ms_a / 1e9 / 157.3:.4f} of peak)  '
          f'critic {ms_c * 1e3:.1f} us ({bench.CRITIC_FLOP_PER_SAMPLE * n / ms_c / 1e9 / 157.3:.4f})')
