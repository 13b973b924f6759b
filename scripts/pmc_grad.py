"""Launches only the fused grad kernels (N = 4096*256) a few times: target of rocprofv3 --pmc."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib

lib = _lib.load()
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.check(lib.tonic_set_tuning(b'grad_variant', waves), 'tuning')
O, A, n = 17, 6, 4096 * 256
g = torch.Generator(device='cuda'); g.manual_seed(0)
P = lib.tonic_ppo_actor_param_count(O, A)
Pc = lib.tonic_v_critic_param_count(O)
params = torch.randn(P, device='cuda', generator=g) * 0.1
cparams = torch.randn(Pc, device='cuda', generator=g) * 0.1
obs = torch.randn(n, O, device='cuda', generator=g)
act = torch.randn(n, A, device='cuda', generator=g).clamp(-1, 1)
adv = torch.randn(n, device='cuda', generator=g)
logp = torch.randn(n, device='cuda', generator=g) * 0.1 - 6
ret = torch.randn(n, device='cuda', generator=g)
stats = torch.tensor([0., 1., 0., 0.], device='cuda')
mean, std = torch.zeros(O, device='cuda'), torch.ones(O, device='cuda')
out = torch.zeros(P + 8, device='cuda')
outc = torch.zeros(Pc + 8, device='cuda')
ws = torch.empty(lib.tonic_mlp64_grad_workspace_bytes(n, P), dtype=torch.uint8, device='cuda')
p = _lib.ptr
for _ in range(reps):
    _lib.check(lib.tonic_ppo_actor_grad(p(params), p(obs), p(act), p(adv), p(stats), p(logp), p(out),
                                        n, O, A, 0.2, 0.0, None, 0, p(ws), ws.numel(), None), 'actor')
    _lib.check(lib.tonic_value_regression_grad(p(cparams), p(mean), p(std), 0.0, p(obs), p(ret), p(outc),
                                               n, O, 0, p(ws), ws.numel(), None), 'critic')
torch.cuda.synchronize()
print('done', float(out.abs().sum()), float(outc.abs().sum()))
