"""Times the three builds of the fused grad kernels (tonic_set_tuning "grad_variant": 0 = 32x32x2 fp32,
1 = 16x16x4 fp32, 2 = 16x16x4 with bf16x3 hidden-layer products) at the benchmark size and prints how far
their gradient sums are from variant 1's."""
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
lib = _lib.load()
replay, actor, critic = agent.replay, agent.actor_updater, agent.critic_updater
b = replay.buffers
n = bench.T * bench.W
obs, act, adv, logp, ret = (replays.flatten_batch(b[k]) for k in
                            ('observations', 'actions', 'advantages', 'log_probs', 'returns'))


def actor_grad():
    actor.enqueue_grad(obs, act, adv, replay.adv_stats, logp)


def critic_grad():
    critic.enqueue_grad(obs, ret)


sums = {}
for variant in (1, 0, 2, 3, 2, 3):
    _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
    actor._workspace_for(n), critic._workspace_for(n)
    ms_a, ms_c = bench.time_events(actor_grad, 20), bench.time_events(critic_grad, 20)
    torch.cuda.synchronize()
    sums[variant] = (actor.grad_sums.clone().double(), critic.grad_sums.clone().double())
    print(f'variant {variant}: actor {ms_a * 1e3:.1f} us ({bench.ACTOR_FLOP_PER_SAMPLE * n / ms_a / 1e9 / 157.3:.4f} '
          f'of the fp32 peak)  critic {ms_c * 1e3:.1f} us ({bench.CRITIC_FLOP_PER_SAMPLE * n / ms_c / 1e9 / 157.3:.4f})')
for variant in (0, 2, 3):
    for name, got, want in zip(('actor', 'critic'), sums[variant], sums[1]):
        scale = want.abs().max()
        print(f'variant {variant} vs 1, {name}: max |diff| / max |grad| = {((got - want).abs().max() / scale).item():.3e}, '
              f'stats {got[-8:].tolist()[:3]} vs {want[-8:].tolist()[:3]}')
_lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')
