"""Per-shape timing of the small-batch GEMM building block on the SAC / TD3 shapes (hipGraph of 50
back-to-back launches, so the number is the kernel, not the Python launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load(); p = _lib.ptr
shapes = [('NT', 1024, 256, 119), ('NT', 1024, 256, 111), ('NT', 1024, 256, 256), ('NT', 1024, 8, 256),
          ('NT', 1024, 1, 256), ('NN', 1024, 256, 256), ('NN', 1024, 119, 256), ('NN', 1024, 256, 8),
          ('TN', 256, 119, 1024), ('TN', 256, 256, 1024), ('TN', 8, 256, 1024), ('TN', 1, 256, 1024)]
for mode, M, N, K in shapes:
    a = torch.randn((M, K) if mode[0] == 'N' else (K, M), device='cuda')
    b = torch.randn((N, K) if mode[1] == 'T' else (K, N), device='cuda')
    c = torch.zeros(M, N, device='cuda')
    def f():
        for _ in range(50):
            _lib.check(lib.tonic_gemm_f32(mode.encode(), p(a), p(b), p(c), None, None, None, M, N, K,
                                          a.shape[1], b.shape[1], N, 0, 0, 1.0, _lib.current_stream()), 'g')
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 100 * 1e3
    print(f'{mode} M={M:5d} N={N:4d} K={K:5d}: {us:6.2f} us  {2.0 * M * N * K / us / 1e6:7.2f} TFLOP/s')
