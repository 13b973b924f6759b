"""Times the fused grad kernels per grad_variant (N = 4096 x 256, O = 17, A = 6): `reps` back-to-back launches
between two events, best of three rounds; prints the error of every variant's gradient sums against variant 1
(fp32 MFMA) beside it.  usage: grad_variant_timing.py [variants, e.g. 1,3,4] [reps]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib

lib = _lib.load()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '1,3,4').split(',')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
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


ref = None
for v in variants:
    _lib.check(lib.tonic_set_tuning(b'grad_variant', v), 'tuning')
    actor(); critic(); torch.cuda.synchronize()
    ga, gc = out.cpu().numpy().astype(np.float64), outc.cpu().numpy().astype(np.float64)
    if ref is None:
        ref = (ga, gc)
    ea = np.abs(ga[:P] - ref[0][:P]).max() / np.abs(ref[0][:P]).max()
    ec = np.abs(gc[:Pc] - ref[1][:Pc]).max() / np.abs(ref[1][:Pc]).max()
    print(f'variant {v}: actor {timed(actor):7.1f} us  critic {timed(critic):7.1f} us   '
          f'max |diff| / max |grad| vs variant {variants[0]}: actor {ea:.2e} critic {ec:.2e}  '
          f'finite {np.isfinite(ga).all() and np.isfinite(gc).all()}', flush=True)
_lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')
