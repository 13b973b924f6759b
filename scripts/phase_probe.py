import os, sys
import numpy as np, torch
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
ph = torch.zeros(8 * 12, dtype=torch.int64, device='cuda')
p = _lib.ptr
for _ in range(2):
    _lib.check(lib.tonic_debug_grad16_phases(p(params), p(obs), p(act), p(adv), p(stats), p(logp), n, O, A,
                                             p(ws), ws.numel(), p(ph), None), 'probe')
torch.cuda.synchronize()
m = ph.cpu().numpy().reshape(8, 12).astype(np.float64) / 32   # per tile (32 tiles per wave)
names = ['in+L1', 'tanh1', 'L2', 'tanh2', 'head+loss', 'dz2+scat', 'dh1', 'dW3+gath', 'dz1+scat', 'dW2', 'dW1', 'loop']
print('cycles per 16-sample tile (s_memtime ticks), wave 0 / wave 4 / mean over 8 waves')
for k, nm in enumerate(names):
    print(f'{nm:10s} {m[0, k]:8.0f} {m[4, k]:8.0f} {m[:, k].mean():8.0f}')
print('total     ', m[0].sum(), m[4].sum(), m.sum(1).mean())
