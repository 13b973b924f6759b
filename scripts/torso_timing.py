"""One PPO learner iteration (actor grad + Adam + critic grad + Adam) at N = 4096 x 256 for torsos other than the
default, on the layer-by-layer HIP path (tonic_*_torso) and as stock torch operators (TONIC_AMD_TORSO_STOCK=1);
the default torso's fused kernels beside them.  usage: torso_timing.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tonic_amd
import tonic_amd.torch as tt
from tonic_amd.environments import Box
from tonic_amd.utils import logger

logger.get_current_logger().store = lambda *a, **k: None
O, A, n = 17, 6, 4096 * 256


def iteration_ms(sizes, activation, stock):
    os.environ['TONIC_AMD_TORSO_STOCK'] = '1' if stock else '0'
    act = getattr(torch.nn, activation)
    model = tt.models.ActorCritic(
        actor=tt.models.Actor(encoder=tt.models.ObservationEncoder(), torso=tt.models.MLP(sizes, act),
                              head=tt.models.DetachedScaleGaussianPolicyHead()),
        critic=tt.models.Critic(encoder=tt.models.ObservationEncoder(), torso=tt.models.MLP(sizes, act),
                                head=tt.models.ValueHead()),
        observation_normalizer=tt.normalizers.MeanStd())
    agent = tt.agents.PPO(model=model, replay=tonic_amd.replays.Segment(size=4096, batch_iterations=80))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    obs = torch.randn(n, O, device='cuda', generator=g)
    actions = torch.randn(n, A, device='cuda', generator=g).clamp(-1, 1)
    adv = torch.randn(n, device='cuda', generator=g)
    logp = torch.randn(n, device='cuda', generator=g) * 0.1 - 6
    ret = torch.randn(n, device='cuda', generator=g)
    stats = torch.tensor([0., 1., 0., 0.], device='cuda')
    infos = torch.zeros(2, 8, device='cuda')
    actor, critic = agent.actor_updater, agent.critic_updater
    kind = 'stock' if actor.stock else 'hip-torso' if actor.torso is not None else 'fused'

    def one():
        actor.reset_stop()
        actor.enqueue_grad(obs, actions, adv, stats, logp)
        actor.enqueue_step(n, stats, infos[0])
        critic.enqueue_grad(obs, ret)
        critic.enqueue_step(n, infos[1])
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    agent.close()
    return kind, ms


for sizes, activation in (((64, 64), 'Tanh'), ((256, 256), 'ReLU'), ((96, 48, 32), 'Tanh'), ((64, 64), 'ReLU'),
                          ((384, 300), 'ReLU')):
    row = []
    for stock in ((False,) if (sizes, activation) == ((64, 64), 'Tanh') else (False, True)):
        kind, ms = iteration_ms(sizes, activation, stock)
        row.append(f'{kind} {ms:8.2f} ms')
    print(sizes, activation, ' | '.join(row), flush=True)
