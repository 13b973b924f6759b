"""Where does the per-step time of the packed collect kernel go at T = 4096?  Variants: inputs
walking through HBM or pinned to step 0, Segment rows walking or pinned to row 0."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load(); p = _lib.ptr
O, A, W, T = 17, 6, 256, 4096
P = lib.tonic_ppo_actor_param_count(O, A)
params = torch.randn(P, device='cuda') * 0.1
packed = torch.empty(lib.tonic_ppo_packed_actor_floats(O, A), device='cuda')
_lib.check(lib.tonic_ppo_pack_actor(p(params), p(packed), O, A, None), 'pack')
obs = torch.randn(T + 1, W, O, device='cuda'); eps = torch.randn(T, W, A, device='cuda')
rew = torch.randn(T, W, device='cuda'); z = torch.zeros(T, W, device='cuda')
seg = {k: torch.zeros(T, W, d, device='cuda') for k, d in (('o', O), ('a', A), ('n', O))}
sv = {k: torch.zeros(T, W, device='cuda') for k in 'rstl'}
sums = torch.zeros(2 * O, device='cuda')


def run(walk_inputs, walk_rows, record=True):
    def f():
        for t in range(T):
            ti = t if walk_inputs else 0
            _lib.check(lib.tonic_ppo_collect_step_packed(
                p(packed), p(obs[ti]), p(eps[ti]), p(obs[ti + 1]), p(rew[ti]), p(z[ti]), p(z[ti]),
                p(seg['o']), p(seg['a']), p(seg['n']), p(sv['r']), p(sv['s']), p(sv['t']), p(sv['l']),
                p(sums) if record else None, None, t if walk_rows else 0, W, O, A,
                _lib.current_stream()), 'c')
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 3 / T * 1e3


for wi in (True, False):
    for wr in (True, False):
        print(f'inputs {"walk" if wi else "pinned"}, rows {"walk" if wr else "pinned"}: '
              f'{run(wi, wr):.2f} us per step')
print(f'inputs walk, rows walk, no record: {run(True, True, False):.2f} us per step')
