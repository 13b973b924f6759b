"""Times the fused grad kernels for several phase-skew settings of the 16x16x4 variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load()
O, A, n = 17, 6, 4096 * 256
g = torch.Generator(device='cuda'); g.manual_seed(0)
P, Pc = lib.tonic_ppo_actor_param_count(O, A), lib.tonic_v_critic_param_count(O)
params = torch.randn(P, device='cuda', generator=g) * 0.1
cparams = torch.randn(Pc, device='cuda', generator=g) * 0.1
obs = torch.randn(n, O, device='cuda', generator=g)
act = torch.randn(n, A, device='cuda', generator=g).clamp(-1, 1)
adv = torch.randn(n, device='cuda', generator=g)
logp = torch.randn(n, device='cuda', generator=g) * 0.1 - 6
ret = torch.randn(n, device='cuda', generator=g)
stats = torch.tensor([0., 1., 0., 0.], device='cuda')
mean, std = torch.zeros(O, device='cuda'), torch.ones(O, device='cuda')
out, outc = torch.zeros(P + 8, device='cuda'), torch.zeros(Pc + 8, device='cuda')
ws = torch.empty(lib.tonic_mlp64_grad_workspace_bytes(n, P), dtype=torch.uint8, device='cuda')
p = _lib.ptr
def actor():
    _lib.check(lib.tonic_ppo_actor_grad(p(params), p(obs), p(act), p(adv), p(stats), p(logp), p(out), n, O, A, 0.2, 0.0, None, p(ws), ws.numel(), None), 'a')
def critic():
    _lib.check(lib.tonic_value_regression_grad(p(cparams), p(mean), p(std), p(obs), p(ret), p(outc), n, O, p(ws), ws.numel(), None), 'c')
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for rnd in range(2):
    for variant, skews in ((0, [0]), (1, [0, 1, 2, 3, 4, 6, 8])):
        _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 't')
        for sk in skews:
            _lib.check(lib.tonic_set_tuning(b'grad_skew', sk), 't')
            print(f'round {rnd} variant {variant} skew {sk}: actor {timeit(actor):.4f} ms critic {timeit(critic):.4f} ms', flush=True)
