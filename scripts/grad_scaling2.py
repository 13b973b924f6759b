"""Fixed cost of a fused grad launch: times n = 32768 x {1, 2, 4, 8, 16, 32} samples (1 .. 32 tiles per wave)."""
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
obs, act, adv, logp, ret = (replays.flatten_batch(b[k]) for k in
                            ('observations', 'actions', 'advantages', 'log_probs', 'returns'))
stream = _lib.current_stream()
mean, std = critic.norm_tensors()
for tiles in (1, 2, 4, 8, 16, 32):
    n = 32768 * tiles
    ws = actor._workspace_for(n)
    wsc = critic._workspace_for(n)

    def actor_grad():
        _lib.check(lib.tonic_ppo_actor_grad(
            p(actor.flat.flat), p(obs), p(act), p(adv), p(replay.adv_stats), p(logp),
            p(actor.grad_sums), n, bench.O, bench.A, 0.2, 0.0, None, 0, p(ws), ws.numel(), stream), 'actor')

    def critic_grad():
        _lib.check(lib.tonic_value_regression_grad(
            p(critic.flat.flat), p(mean), p(std), 0.0, p(obs), p(ret), p(critic.grad_sums), n, bench.O, 0,
            p(wsc), wsc.numel(), stream), 'critic')

    ms_a, ms_c = bench.time_events(actor_grad, 20), bench.time_events(critic_grad, 20)
    print(f'{tiles:3d} tiles per wave: actor {ms_a * 1e3:7.1f} us  critic {ms_c * 1e3:7.1f} us')
