import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load(); p = _lib.ptr
O, A, W, T = 17, 6, 256, 64
P = lib.tonic_ppo_actor_param_count(O, A)
params = torch.randn(P, device='cuda') * 0.1
packed = torch.empty(lib.tonic_ppo_packed_actor_floats(O, A), device='cuda')
_lib.check(lib.tonic_ppo_pack_actor(p(params), p(packed), O, A, None), 'pack')
obs = torch.randn(T + 1, W, O, device='cuda'); eps = torch.randn(T, W, A, device='cuda')
rew = torch.randn(T, W, device='cuda'); z = torch.zeros(T, W, device='cuda')
seg = {k: torch.zeros(T, W, d, device='cuda') for k, d in (('o', O), ('a', A), ('n', O))}
sv = {k: torch.zeros(T, W, device='cuda') for k in 'rstl'}
sums = torch.zeros(2 * O, device='cuda')
def run(use_sums, Wn=W):
    def f():
        for t in range(T):
            _lib.check(lib.tonic_ppo_collect_step_packed(p(packed), p(obs[t]), p(eps[t]), p(obs[t + 1]), p(rew[t]), p(z[t]), p(z[t]),
                p(seg['o']), p(seg['a']), p(seg['n']), p(sv['r']), p(sv['s']), p(sv['t']), p(sv['l']),
                p(sums) if use_sums else None, None, t, Wn, O, A, _lib.current_stream()), 'c')
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 / T * 1e3
print('us per step, with record   :', run(True))
print('us per step, without record:', run(False))
print('us per step, W=16 no record:', run(False, 16))


def empty_node_floor():
    """A graph of T dependent near-empty launches (tonic_polyak_update on 64 floats)."""
    a, b = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')

    def f():
        for _ in range(T):
            _lib.check(lib.tonic_polyak_update(p(a), p(b), 64, 0.5, _lib.current_stream()), 'p')
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 / T * 1e3


print('us per node, near-empty kernel chain:', empty_node_floor())
