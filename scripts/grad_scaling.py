"""Fixed cost vs per-sample cost of the fused grad kernels: time at n = 2^20 ... 2^15."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load(); p = _lib.ptr
O, A, N = 17, 6, 1 << 20
g = torch.Generator(device='cuda'); g.manual_seed(0)
P = lib.tonic_ppo_actor_param_count(O, A); Pc = lib.tonic_v_critic_param_count(O)
params = torch.randn(P, device='cuda', generator=g) * 0.1
cparams = torch.randn(Pc, device='cuda', generator=g) * 0.1
obs = torch.randn(N, O, device='cuda', generator=g)
act = torch.randn(N, A, device='cuda', generator=g).clamp(-1, 1)
adv = torch.randn(N, device='cuda', generator=g)
logp = torch.randn(N, device='cuda', generator=g) * 0.1 - 6
ret = torch.randn(N, device='cuda', generator=g)
st = torch.tensor([0., 1., 0., 0.], device='cuda')
mean, std = torch.zeros(O, device='cuda'), torch.ones(O, device='cuda')
out, outc = torch.zeros(P + 8, device='cuda'), torch.zeros(Pc + 8, device='cuda')
ws = torch.empty(lib.tonic_mlp64_grad_workspace_bytes(N, P), dtype=torch.uint8, device='cuda')


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n in (1 << 20, 1 << 19, 1 << 18, 1 << 17, 1 << 16, 1 << 15):
    a = timed(lambda: _lib.check(lib.tonic_ppo_actor_grad(
        p(params), p(obs), p(act), p(adv), p(st), p(logp), p(out), n, O, A, 0.2, 0.0, None, 0, p(ws),
        ws.numel(), None), 'a'))
    c = timed(lambda: _lib.check(lib.tonic_value_regression_grad(
        p(cparams), p(mean), p(std), 0.0, p(obs), p(ret), p(outc), n, O, 0, p(ws), ws.numel(), None), 'c'))
    print(f'n={n:8d}  actor {a:8.1f} us  critic {c:8.1f} us   per 16-sample tile and wave: '
          f'{a * 2.4e3 / max(n / 16 / 2048, 1):8.0f} / {c * 2.4e3 / max(n / 16 / 2048, 1):8.0f} cycles')
