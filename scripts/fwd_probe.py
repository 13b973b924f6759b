"""Times tonic_policy_forward (fused actor forward + sampling) at SAC shapes inside a hipGraph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load(); p = _lib.ptr
B, O, H, A = 1024, 111, 256, 8
P = lib.tonic_mlp_actor_param_count(O, H, A, 2)
params = torch.randn(P, device='cuda') * 0.05
obs = torch.randn(B, O, device='cuda'); eps = torch.randn(B, A, device='cuda')
act = torch.zeros(B, A, device='cuda')
ws = torch.empty(lib.tonic_offpolicy_workspace_bytes(B, O, A, H), dtype=torch.uint8, device='cuda')
def f():
    for _ in range(50):
        _lib.check(lib.tonic_policy_forward(p(params), p(obs), p(eps), p(act), 1, B, O, H, A, p(ws), ws.numel(),
                                            _lib.current_stream()), 'f')
f(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): f()
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
print(f'policy forward (mlp forward + sample): {s.elapsed_time(e) / 100 * 1e3:.2f} us per call')
