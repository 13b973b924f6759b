"""Time of one PPO iteration (actor grad + critic grad) on the layer-by-layer path for shapes beyond
the fused kernels, at the benchmark's batch size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from tonic_amd import _lib                      # noqa: E402

lib, p = _lib.load(), _lib.ptr
SHAPES = ((111, 8, 1 << 20), (376, 17, 1 << 20), (17, 6, 1 << 20))
if os.environ.get("WIDE_ONLY"):
    SHAPES = SHAPES[:1]
for O, A, n in SHAPES:
    Pa, Pc = lib.tonic_ppo_actor_param_count(O, A), lib.tonic_v_critic_param_count(O)
    gen = torch.Generator(device='cuda').manual_seed(0)
    actor = torch.randn(Pa, device='cuda', generator=gen) * 0.1
    critic = torch.randn(Pc, device='cuda', generator=gen) * 0.1
    obs = torch.randn(n, O, device='cuda', generator=gen)
    act = torch.randn(n, A, device='cuda', generator=gen).clamp(-1, 1)
    adv = torch.randn(n, device='cuda', generator=gen)
    lp = torch.randn(n, device='cuda', generator=gen) * 0.1 - A
    ret = torch.randn(n, device='cuda', generator=gen)
    stats = torch.tensor([0., 1., 0., 0.], device='cuda')
    mean, std = torch.zeros(O, device='cuda'), torch.ones(O, device='cuda')
    ga, gc = torch.zeros(Pa + 8, device='cuda'), torch.zeros(Pc + 8, device='cuda')
    wa = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, A, 1), dtype=torch.uint8, device='cuda')
    wc = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, 1, 0), dtype=torch.uint8, device='cuda')

    def actor_grad():
        _lib.check(lib.tonic_ppo_actor_grad(p(actor), p(obs), p(act), p(adv), p(stats), p(lp), p(ga),
                                            n, O, A, 0.2, 0.0, None, 0, p(wa), wa.numel(), None), 'a')

    def critic_grad():
        _lib.check(lib.tonic_value_regression_grad(p(critic), p(mean), p(std), 0.0, p(obs), p(ret),
                                                   p(gc), n, O, 0, p(wc), wc.numel(), None), 'c')
    ms_a, ms_c = bench.time_events(actor_grad, 5), bench.time_events(critic_grad, 5)
    flop_a = 2 * (O * 64 + 4096 + 64 * A) * 3 - 2 * O * 64      # fwd + dW + dX (no dX for layer 1)
    print(f'O={O} A={A} N={n}: actor grad {ms_a:.3f} ms, critic grad {ms_c:.3f} ms, '
          f'workspace {wa.numel() / 1e9:.2f} GB, actor {flop_a * n / ms_a / 1e9:.1f} TFLOP/s')
