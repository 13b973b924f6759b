"""Workload for the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass):
the GAE scan at T=4096, W=65536 (1 chunk: 5.4 GB read, 2.1 GB written — a known byte count that
calibrates the counters for this access pattern) and the fused grad kernels at N = 1 048 576."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib

lib = _lib.load()
p = _lib.ptr
g = torch.Generator(device='cuda'); g.manual_seed(0)

T, W = 4096, 65536
arrs = [torch.randn(T, W, device='cuda', generator=g) for _ in range(2)]          # next_values, rewards
resets = (torch.rand(T, W, device='cuda', generator=g) < 1e-3).float()
terms = resets * (torch.rand(T, W, device='cuda', generator=g) < 0.5).float()
values = torch.randn(T, W, device='cuda', generator=g)
ret, adv = torch.empty(T, W, device='cuda'), torch.empty(T, W, device='cuda')
stats = torch.zeros(4, device='cuda')
ws = torch.empty(max(lib.tonic_gae_workspace_bytes(T, W, 1), 16), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.tonic_gae_lambda_returns(
        p(arrs[0]), p(arrs[1]), p(resets), p(terms), p(values), p(ret), p(adv), p(stats), None,
        T, W, 0.99, 0.97, 1, p(ws), ws.numel(), None), 'gae')
torch.cuda.synchronize()
del arrs, resets, terms, values, ret, adv

# the bit-exact scan at the metric's own size (gae_stream16_kernel: T = 4096, W = 256; 29.4 MB)
T2, W2 = 4096, 256
small = [torch.randn(T2, W2, device='cuda', generator=g) for _ in range(3)]       # next_values, rewards, values
rs2 = (torch.rand(T2, W2, device='cuda', generator=g) < 1e-3).float()
tm2 = rs2 * (torch.rand(T2, W2, device='cuda', generator=g) < 0.5).float()
ret2, adv2 = torch.empty(T2, W2, device='cuda'), torch.empty(T2, W2, device='cuda')
ws2 = torch.empty(max(lib.tonic_gae_workspace_bytes(T2, W2, 1), 16), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.tonic_gae_lambda_returns(
        p(small[0]), p(small[1]), p(rs2), p(tm2), p(small[2]), p(ret2), p(adv2), p(stats), None,
        T2, W2, 0.99, 0.97, 1, p(ws2), ws2.numel(), None), 'gae small')
torch.cuda.synchronize()
# ... and the Segment's default there since round 6: the one-pass form (gae_onepass_kernel, chunks = 0)
ws3 = torch.empty(max(lib.tonic_gae_workspace_bytes(T2, W2, 0), 16), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.tonic_gae_lambda_returns(
        p(small[0]), p(small[1]), p(rs2), p(tm2), p(small[2]), p(ret2), p(adv2), p(stats), None,
        T2, W2, 0.99, 0.97, 0, p(ws3), ws3.numel(), None), 'gae small, one pass')
torch.cuda.synchronize()

O, A, n = 17, 6, 4096 * 256
P = lib.tonic_ppo_actor_param_count(O, A)
Pc = lib.tonic_v_critic_param_count(O)
params = torch.randn(P, device='cuda', generator=g) * 0.1
cparams = torch.randn(Pc, device='cuda', generator=g) * 0.1
obs = torch.randn(n, O, device='cuda', generator=g)
act = torch.randn(n, A, device='cuda', generator=g).clamp(-1, 1)
advn = torch.randn(n, device='cuda', generator=g)
logp = torch.randn(n, device='cuda', generator=g) * 0.1 - 6
rets = torch.randn(n, device='cuda', generator=g)
st = torch.tensor([0., 1., 0., 0.], device='cuda')
mean, std = torch.zeros(O, device='cuda'), torch.ones(O, device='cuda')
out, outc = torch.zeros(P + 8, device='cuda'), torch.zeros(Pc + 8, device='cuda')
ws = torch.empty(lib.tonic_mlp64_grad_workspace_bytes(n, P), dtype=torch.uint8, device='cuda')
for _ in range(3):
    _lib.check(lib.tonic_ppo_actor_grad(p(params), p(obs), p(act), p(advn), p(st), p(logp), p(out),
                                        n, O, A, 0.2, 0.0, None, 0, p(ws), ws.numel(), None), 'actor')
    _lib.check(lib.tonic_value_regression_grad(p(cparams), p(mean), p(std), 0.0, p(obs), p(rets), p(outc),
                                               n, O, 0, p(ws), ws.numel(), None), 'critic')
# the critic's forward over the whole Segment (mlp64_grad16_kernel<..., FWD>: 68 B read + 4 B written per row)
vals = torch.empty(n, device='cuda')
for _ in range(3):
    _lib.check(lib.tonic_value_forward(p(cparams), p(mean), p(std), 0.0, p(obs), p(vals), n, O, None), 'values')
torch.cuda.synchronize()
print('done')
