"""Times the shipped fused grad kernels (N = 4096 x 256, O = 17, A = 6) under the timing-only knobs of
tonic_set_tuning: grad_prio (wave priorities against the issue arbiter's age order) and grad_skew (start
offset of the second-dispatched half of a workgroup); checks that the gradient sums do not change by a bit.
usage: grad_knobs_timing.py [reps]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
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


def actor():
    _lib.check(lib.tonic_ppo_actor_grad(p(params), p(obs), p(act), p(adv), p(stats), p(logp), p(out),
                                        n, O, A, 0.2, 0.0, None, 0, p(ws), ws.numel(), None), 'actor')


def critic():
    _lib.check(lib.tonic_value_regression_grad(p(cparams), p(mean), p(std), 0.0, p(obs), p(ret), p(outc),
                                               n, O, 0, p(ws), ws.numel(), None), 'critic')


def timed(fn):
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3


for _ in range(40):           # the device's clock follows its load: warm it
    actor(); critic()
torch.cuda.synchronize()
ref = None
for prio, skew in [(int(a), int(b)) for a, b in (c.split(':') for c in (sys.argv[2] if len(sys.argv) > 2 else '0:0,1:0,2:0,0:1,0:4,2:1,0:0').split(','))]:
    _lib.check(lib.tonic_set_tuning(b'grad_prio', prio), 'tuning')
    _lib.check(lib.tonic_set_tuning(b'grad_skew', skew), 'tuning')
    actor(); critic(); torch.cuda.synchronize()
    bits = (out.cpu().numpy().tobytes(), outc.cpu().numpy().tobytes())
    ref = ref or bits
    print(f'grad_prio {prio} grad_skew {skew:2d}: actor {timed(actor):7.1f} us  critic {timed(critic):7.1f} us  '
          f'same bits: {bits == ref}', flush=True)
_lib.check(lib.tonic_set_tuning(b'grad_prio', 2), 'tuning')
_lib.check(lib.tonic_set_tuning(b'grad_skew', 0), 'tuning')
