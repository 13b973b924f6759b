"""Per-phase cycle totals of the fused actor grad kernel (tonic_debug_grad16_phases: s_memtime stamps in the
tile loop of workgroup 0's eight waves, shipped arithmetic), N = 4096 x 256, O = 17, A = 6.  The stamps
themselves cost ~10 % (every stamp pins the schedule); read the SHARES."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib

lib = _lib.load()
O, A, n = 17, 6, 4096 * 256
g = torch.Generator(device='cuda'); g.manual_seed(0)
P = lib.tonic_ppo_actor_param_count(O, A)
params = torch.randn(P, device='cuda', generator=g) * 0.1
obs = torch.randn(n, O, device='cuda', generator=g)
act = torch.randn(n, A, device='cuda', generator=g).clamp(-1, 1)
adv = torch.randn(n, device='cuda', generator=g)
logp = torch.randn(n, device='cuda', generator=g) * 0.1 - 6
stats = torch.tensor([0., 1., 0., 0.], device='cuda')
ws = torch.empty(lib.tonic_mlp64_grad_workspace_bytes(n, P), dtype=torch.uint8, device='cuda')
cycles = torch.zeros(8 * 12, dtype=torch.int64, device='cuda')
p = _lib.ptr
for _ in range(3):
    _lib.check(lib.tonic_debug_grad16_phases(p(params), p(obs), p(act), p(adv), p(stats), p(logp), n, O, A,
                                             p(ws), ws.numel(), p(cycles), None), 'phases')
torch.cuda.synchronize()
c = cycles.cpu().numpy().reshape(8, 12).astype(np.float64)
tiles = n / 16 / (256 * 8)
names = ['layer 1 issued', 'tanh 1', 'layer 2 chain', 'tanh 2', 'head + loss', 'dz2 + scatters', 'backward chain',
         'dW3 + dz2^T gathers', 'dz1 + scatters', 'dW2', 'dW1', 'loop / prefetch']
per_tile = c.mean(0) / tiles
for name, v in zip(names, per_tile):
    print(f'{name:22s} {v:8.0f} cycles per tile and wave  {100 * v / per_tile.sum():5.1f} %')
print(f'{"sum":22s} {per_tile.sum():8.0f}   (waves: {", ".join("%.0f" % (w / tiles) for w in c.sum(1))})')
